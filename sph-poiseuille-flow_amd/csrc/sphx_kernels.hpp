// sphx_kernels.hpp -- the device-resident SPH step for MI355X (gfx950): HIP kernels.
//
// One time step of SPH_Poiseuille.m:250-292 (density_correction -> viscous_force + gravity ->
// transport_correction -> verlet_time_step -> integration_verlet -> periodic wrap -> neighbour
// rebuild) is four neighbour passes with no host round trip,
//     A k_density (list + density)  ->  B k_kgc  ->  CD k_forces  ->  E k_continuity (+ clock)
// followed on every K-th step by the re-binning chain k_clock_scan -> k_scatter -> k_reorder.  On small channels pass E
// of a step and pass A of the next one share a launch (k_continuity_density): 3 launches per step.  Design (long form
// in DESIGN.md):
//   * particles are sorted by cell, cell id = cx*ncy + cy (y fastest): the 3x3 neighbourhood of a cell is
//     three contiguous index ranges and an x-slab of the channel is one contiguous range; everything a
//     pass gathers per neighbour is a 16- or 32-byte record (double2 pos / vel, double4 {Vol,p,rho_h,rho},
//     double4 B) because the passes are bound by the number of scattered loads, not by bytes;
//   * in x the cells tile the period exactly (ncx = floor(DL/(2h+skin)), width >= 2h+skin), so the wrapped
//     3x3 sweep with a min-image dx sees every neighbour and the reference's ghost entries +
//     seen_neighbor (mex/sph_neighbor_search_mex.c:282-295,342,383) are not needed; the accepted set is
//     the same {1e-24 < r^2 < (2h)^2} (:368).  A slab of a multi-GPU run uses the same kernels on an
//     open (non-periodic) window of columns;
//   * cell skin: between two re-binnings the layout is frozen, sweeps are centred on the cell a particle
//     was binned into, and the clock stops the loop (or, large channels and slabs: re-bins by itself) before any
//     particle is further than skin/2 from where it was binned -- see Clock::drift, FluidSet::cell / posb;
//   * pass A records the accepted neighbours (fluid first, then wall) of the step in a lane-major index list;
//     the other three passes walk that list (all lanes busy, no cut-off branch, no cell lookups) -- the
//     geometry is frozen for the step exactly as in the reference, which reuses dx,dy,r,W,dW of the pair
//     list built at the end of the previous step.  The first step after a re-binning also records every
//     pair within 2h+skin (superset list); the other steps of the cycle walk that instead of the cells;
//   * every pair sum is a per-particle gather (each fluid-fluid update of mex/sph_physics_mex.c is
//     symmetric under i<->j): no atomics in the physics, bitwise reproducible run to run;
//   * LPP lanes of a wavefront share one particle's ring and combine with __shfl_xor (wave64): 32 / 16 at a few thousand
//     particles (compact kernels: a pass is a chain of dependent memory round trips, every pass requests its
//     own-particle data, its list row count and its first list rows before it waits for the run flag), 8 / 4 / 2 above
//     (the "_w" forms: list entries three rows ahead, fluid / wall loops, LDS tiles of the neighbourhood);
//   * dt, t, the step counter, the particle count and the stop test live in a device-side clock so
//     steps can be captured into a hipGraph and replayed.  Small channels: the clock update rides in a tail workgroup of
//     the last launch of a step (continuity_tail).  Large channels ("dynamic" contexts) and slabs: the clock also decides
//     when to re-bin and the re-binning kernels of every step skip themselves otherwise (slot_active, clock_step);
//   * x-slabs (multi-GPU): k_slab_* at the end of this file.
#pragma once
#include <type_traits>

#include "sphx_device.hpp"

namespace sphx {

constexpr int kBlock = 256;
constexpr int kScanBlock = 1024;
constexpr int kBigScanCells = 8192;  // above this the cell scan runs as three multi-block kernels

struct Grid {
    int ncx, ncy, ncells;
    int periodic;         // 1: columns wrap in x (single GPU); 0: open window (slab)
    double DL, half_DL;   // min-image: |dx| > half_DL -> dx -+ DL ; half_DL = +inf in an open window
    double x0, y0, inv_csx, inv_csy;
    double own_lo, own_hi;  // particles with own_lo <= x < own_hi are owned by this context
    // Slabs that re-bin every K-th step: between two re-binnings ownership is frozen, it goes by the column a particle
    // was BINNED into (own_c0 <= column < own_c1), not by where it has drifted since
    int own_by_cell, own_c0, own_c1;
};

__device__ __forceinline__ bool owns(const Grid &g, double x, int cell)
{
    if (!g.own_by_cell) return x >= g.own_lo && x < g.own_hi;
    const int cx = cell / g.ncy;
    return cx >= g.own_c0 && cx < g.own_c1;
}

struct Phys {
    KernelConst kc;
    double rho0, inv_sigma0, mu, p0, c_f, g, tc, nu, DL, DH, w0;
    double dt_viscous, dt_body;  // the two step limits that do not depend on the state (next_dt): 0.125 h^2 / nu, 0.25 sqrt(h / |g|)
};

// Device-side clock: replaces the host variables state.t / state.step / dt_step / remain of
// SPH_Poiseuille.m:247-267 so the loop needs no host decisions.
struct Clock {
    double t, dt, dt_last, t_target, t_end, vmax;
    long long step, steps_left;  // steps_left < 0: unlimited
    int run[2];                  // run[q]: the step slot of parity q executes
    int status;
    int n;                       // particles currently held (fluid, incl. slab halo copies)
    double drift;                // largest distance of any particle from where it was when the grid was built
    int need_rebuild;            // 1: drift exceeded half the cell skin -> the loop stopped, host must re-bin
    int n_rebins;                // dynamic contexts / skinned slabs: re-binnings decided by the clock so far (statistics)
    // "dynamic" contexts (large channels): the device itself decides when to re-bin -- at the K-th step since the
    // last re-binning or as soon as the drift exceeds skin/2 -- and the re-binning kernels of every step skip
    // themselves unless rebuild_now is set; no host round trip, need_rebuild is never raised
    int fresh;                   // 1: the grid has just been rebuilt (pass A sweeps the cells and records the superset list)
    int rebuild_now;             // 1: the re-binning kernels of this step run
    int pos_count;               // steps since the last re-binning
    int n_drift_rebuilds;        // re-binnings triggered by the drift bound (statistics)
    // Whenever the loop stops (step budget used up, target reached, status, stale grid) the clock is also written to
    // `pub`, a copy in pinned host memory: the host then needs no device-to-host copy to learn how a batch ended.
    // seq counts the batches armed (k_prepare), so the host can tell a fresh copy from an old one.
    long long seq;
    Clock *pub;
    // Opt-in dual-rate loop (sphx_params::dual_rate, the outer / inner stepping the reference's README describes): a step
    // slot is one OUTER step of n_in inner sub-steps of length dt -- density, KGC, viscous force and transport shift once,
    // pressure / continuity n_in times.  n_in = 1 is the reference's loop (SPH_Poiseuille.m:250-292), the parity path.
    int n_in;
    // pos_count as the step slot of parity q has to see it: written by the clock update of the slot BEFORE it (parity 1 - q,
    // or k_prepare), never by the slot's own.  k_slab_pack3's workgroups take the re-binning decision each by itself while the
    // last one out already advances the clock: a workgroup dispatched late must still read what the others read.
    // (two shorts in the place of a padding word: a Clock lives in registers in the tail workgroups of pass E, whose kernels
    //  sit exactly at their register budget -- the struct must not grow)
    short pos_q[2];
};

// Everything a neighbour pass GATHERS per neighbour is stored as 16- or 32-byte records (position, velocity,
// {Vol, p_half, rho_half, rho}, B): the passes are bound by the number of scattered load instructions (texture
// addresser), not by bytes, and one 16-byte load costs the same address work as one 8-byte load.
struct FluidSet {  // persistent per-particle state, sorted by cell
    double2 *pos, *vel;
    double *drho, *mass;
    int *id;
    int *start;  // [ncells+1] cell ranges of this ordering
    int *cell;   // [n] the cell each slot was BINNED into (positions may since have drifted by < skin/2)
    double2 *posb;  // [n] position at binning time (nullptr: grid rebuilt every step, drift not tracked)
};

struct FluidTmp {
    double2 *posn, *veln;      // end-of-step state, pre-sort order
    double *drhon;
    double4 *a;                // per-step fields {Vol, p_half, rho_half, rho}
    double4 *B;                // per-step KGC matrix {B11, B12, B21, B22}
    double2 *fp, *f;           // outputs of the step: force_prior, force
    double *rho_out, *p_out;
    int *cellid, *count, *perm, *src_of;
    double *vpart;   // per-block max |v|^2 of pass E (owned particles only)
    double *dpart;   // per-block max drift^2 from the binning positions after this step (pass CD)
    int *nl_idx;     // neighbour list, entry m of lane l at nl_idx[m*nl_stride + l]
    int *nl_cnt;     // [nl_stride] entries per lane
    int *flags;      // [1] sticky device status bits (list overflow ...)
    int *tile_sum;   // big scan: per-tile sums / offsets
    int nl_stride, nl_cap;
    // Superset list (contexts that re-bin every K-th step): every pair within 2h + skin at binning time, same
    // lane-major layout.  Built by pass A of the first step after a re-bin; the other steps of the cycle walk it
    // (~1.3x the true neighbours) instead of sweeping cells (~3.7x).
    int *sl_idx, *sl_cnt;
    int sl_cap;
    double sl_rcut2;  // (2h + skin)^2
    int cap;         // particle capacity of every per-particle array (loads below it are always in bounds)
    int n_vpart;     // entries of vpart / dpart (= workgroups of a pass)
    double half_skin;  // < 0: drift not tracked
    double *vol;     // Vol once more, 8 bytes per particle: passes B and E gather nothing else of `a`, and pulling the
                     // 32-byte records through the caches for one double costs them ~15 % (measured the other way
                     // round: 64-byte records made them 25 % slower)
    // Large-channel kernels (LPP <= 8) keep the FLUID entries of both lists as 16-bit index differences k - i (modulo the
    // population on a periodic grid), two rows per 32-bit word: word p of lane l = rows 2p (low half) and 2p+1 at
    // nl_pk[p*nl_stride + l].  The neighbours of a particle lie in the three cell columns around its own: a few thousand
    // slots either way.  Wall entries (rows behind a lane's fluid rows) stay in nl_idx / sl_idx at their row.  Halves the
    // list bytes the passes move (at 6 M particles the lists were 40-60 % of every pass's HBM traffic).  nullptr: compact kernels.
    int *nl_pk, *sl_pk;
    int has_slack;     // 1: the arrays hold markedly more slots than particles (slabs), see beyond_population
    double *seal_out;  // skinned slabs: the tail workgroup of pass E leaves {max |v|, max drift} of the owned particles here
                       // (the input of the step's max all-reduce), see slab_seal_tail
    int lazy_out;      // large-channel kernels: 1 = the output-only fields of a step (force, force_prior, rho, p) are written by the
                       // LAST step of a batch only, 2 = never, see step_outputs_wanted
    int *tmap;         // large-channel kernels: the tile layout of every workgroup (8 ints each, see tile_map_of), written by the
                       // cell sweep at each re-binning
};

// "no value yet" in vpart when pass E carries the clock update in a tail workgroup (see continuity_tail)
constexpr unsigned long long kVpartEmpty = ~0ull;

struct Walls {
    const double2 *pos;
    const double4 *a;  // {Vol, vx, vy, 0}
    const int *id;
    const int *start;    // [ncells+1]
    const int *row_any;  // [ncy] 1 when rows cy-1..cy+1 hold any wall particle
    int n;
};

__device__ __forceinline__ void cell_of(const Grid &g, double x, double y, int &cx, int &cy)
{
    cx = (int)floor((x - g.x0) * g.inv_csx);
    cx = min(max(cx, 0), g.ncx - 1);
    cy = (int)floor((y - g.y0) * g.inv_csy);
    cy = min(max(cy, 0), g.ncy - 1);
}

// the cell a slot is stored in (set when the grid was built); sweeps are centred on it, not on the cell of the
// current position, because between two grid builds particles drift by up to skin/2
__device__ __forceinline__ void binned_cell(const Grid &g, const FluidSet &s, int i, int &cx, int &cy)
{
    const int c = s.cell[i];
    cx = c / g.ncy;
    cy = c - cx * g.ncy;
}

__device__ __forceinline__ double wrap_x(double x, double DL) { return x - floor(x / DL) * DL; }

// minimum-image separation (mex/sph_neighbor_search_mex.c:357-363); a no-op in an open window
__device__ __forceinline__ double min_image(const Grid &g, double dx)
{
    if (dx > g.half_DL) dx -= g.DL;
    else if (dx < -g.half_DL) dx += g.DL;
    return dx;
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Give each XCD one
// contiguous eighth of the particle range instead, so the neighbour columns a block reads are the ones
// the previous block on the same XCD just pulled into that XCD's L2 (measured at 6 M particles without
// this: k_forces fetched 4x its algorithmic bytes from HBM, L2 hit rate 36 %).  Speed only.
__device__ __forceinline__ int xcd_block(int b, int nb)
{
    const int per = nb >> 3;
    return b < (per << 3) ? (b & 7) * per + (b >> 3) : b;
}

// A period of one or two cell columns (DL < 6h): the columns cx-1, cx, cx+1 are not distinct -- the reference meets the
// same particle as a real and as a ghost entry there and de-duplicates with seen_neighbor
// (mex/sph_neighbor_search_mex.c:282-295,342,383); here each distinct column is swept once and the minimum-image fold
// picks the nearest image.
__device__ __forceinline__ bool duplicate_column(const Grid &g, int ox)
{
    return (g.ncx == 1 && ox != 0) || (g.ncx == 2 && ox == 1);
}

// the value of the other lane of a pair (lanes 2k, 2k + 1): one DPP move (quad_perm [1 0 3 2]), no LDS crossbar, no ballot
__device__ __forceinline__ int pair_partner(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }

// Sum over the LPP lanes of a particle, in every lane.  Up to 16 lanes a group lies in one DPP row: two quad permutes
// (lane ^ 1, lane ^ 2), row_half_mirror and row_mirror pair every lane with one that holds the other part -- two v_mov_dpp
// and an add per stage, where __shfl_xor is two ds_bpermute through the LDS crossbar and a wait (a pass of the small
// channels ends with four to seven such sums, four stages each, all on its critical path).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// max over the wave of non-negative values (squares), in every lane: four DPP stages inside the rows, two crossbar stages across
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    v = fmax(v, __shfl_xor(v, 16));
    return fmax(v, __shfl_xor(v, 32));
}
// largest / smallest value of an int over the LPP lanes of a particle, in every lane (same stages as group_sum)
template <int LPP, bool MAX>
__device__ __forceinline__ int group_extreme(int v)
{
    auto pick = [](int a, int b) { return MAX ? max(a, b) : min(a, b); };
    if (LPP >= 2) v = pick(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));
    if (LPP >= 4) v = pick(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));
    if (LPP >= 8) v = pick(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));
    if (LPP >= 16) v = pick(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true));
#pragma unroll
    for (int off = LPP / 2; off >= 16; off >>= 1) v = pick(v, __shfl_xor(v, off));
    return v;
}
template <int LPP>
__device__ __forceinline__ double group_sum(double v)
{
    if (LPP >= 2) v += dpp_f64<0xB1>(v);    // quad_perm [1 0 3 2]
    if (LPP >= 4) v += dpp_f64<0x4E>(v);    // quad_perm [2 3 0 1]
    if (LPP >= 8) v += dpp_f64<0x141>(v);   // row_half_mirror
    if (LPP >= 16) v += dpp_f64<0x140>(v);  // row_mirror
#pragma unroll
    for (int off = LPP / 2; off >= 16; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Visit the candidates of the three cell columns around (cx,cy): body(k).
template <int LPP, typename Body>
__device__ __forceinline__ void sweep(const Grid &g, const int *__restrict__ start, int cx, int cy,
                                      int sub, Body &&body)
{
    const int cylo = max(cy - 1, 0), cyhi = min(cy + 1, g.ncy - 1);
#pragma unroll
    for (int ox = -1; ox <= 1; ++ox) {
        int col = cx + ox;
        if (g.periodic) {
            if (duplicate_column(g, ox)) continue;
            if (col < 0) col += g.ncx;
            else if (col >= g.ncx) col -= g.ncx;
        } else if (col < 0 || col >= g.ncx) {
            continue;
        }
        const int base = col * g.ncy;
        const int lo = start[base + cylo], hi = start[base + cyhi + 1];
        for (int k = lo + sub; k < hi; k += LPP) body(k);
    }
}

__device__ __forceinline__ double next_dt(const Clock &c, const Phys &ph)
{  // SPH_Poiseuille.m:519-527 with remain of :252
    const double remain = fmin(c.t_target - c.t, c.t_end - c.t);
    const double h = ph.kc.h;
    const double dt_acoustic = 0.25 * h / fmax(ph.c_f + c.vmax, 1e-12);
    // (the clock's thread runs this between two steps, with the whole chip waiting: the viscous and the body-force limit -- two
    //  divisions and a square root of constants -- come precomputed, make_phys)
    const double dt_viscous = ph.dt_viscous, dt_body = ph.dt_body;
    if (c.n_in > 1) {
        // outer step by the advection / viscous / body-force scales, at most n_in acoustic steps long; dt is the inner step
        const double dt_adv = 0.25 * h / fmax(c.vmax, 1e-12);
        const double Dt = fmin(fmin(fmin(dt_adv, dt_viscous), fmin(dt_body, remain)), c.n_in * dt_acoustic);
        return fmax(Dt / c.n_in, 1e-12);
    }
    const double dt = fmin(fmin(dt_acoustic, dt_viscous), fmin(dt_body, remain));
    return fmax(dt, 1e-12);
}

__device__ __forceinline__ bool loop_continues(const Clock &c)
{  // while state.t < target_time - 1e-12 (SPH_Poiseuille.m:250) and step budget left
    return (c.t < c.t_target - 1e-12) && (c.steps_left != 0) && (c.status == 0) && (c.need_rebuild == 0);
}

// force, force_prior (pass CD) and rho, p (pass E) are results for the caller: no later step reads them (the next pass A sums
// the density afresh, the next pass CD builds both forces from scratch), and they are 48 of the ~650 bytes a large-channel
// step moves per particle.  Contexts with FluidTmp::lazy_out write them only in the step after which the batch stops --
// the step budget's last step, or the one that reaches the target time -- which is the only state a caller can look at.
// (The clock still holds the state BEFORE this step while its passes run: clock_step comes after pass E.)
__device__ __forceinline__ bool step_outputs_wanted(const Clock *clk, const FluidTmp &t)
{
    if (!t.lazy_out) return true;
    if (t.lazy_out == 2) return false;  // (a slab: nobody can ask for them, sphx_slab_snapshot hands out the state only)
    return clk->steps_left == 1 || !(clk->t + clk->dt < clk->t_target - 1e-12);  // (the test of loop_continues, a step early)
}

// Kernels of the re-binning chain take the slot parity with a flag: bit 1 set = "only when the clock says
// rebuild_now" (dynamic contexts), so the same kernels serve the scheduled and the device-decided re-binning.
constexpr int kOnlyIfRebuild = 2, kOnlyIfNoHistogram = 4;
__device__ __forceinline__ bool slot_active(const Clock *clk, int qf)
{
    if (!clk->run[qf & 1]) return false;
    if ((qf & kOnlyIfNoHistogram) && clk->rebuild_now == 2) return false;  // pass E binned the particles already
    return !(qf & kOnlyIfRebuild) || clk->rebuild_now != 0;
}

// one thread: arm the clock for an advance call.  vmax_in (optional) overrides the stored vmax (slab:
// the all-reduced global value).
__global__ void k_prepare(Clock *clk, Phys ph, double t_target, long long max_steps, int q0,
                          const double *vmax_in)
{
    Clock c = *clk;
    if (vmax_in) c.vmax = *vmax_in;
    c.t_target = fmin(t_target, c.t_end);  // target_time = min(t + output_interval, t_end), SPH_Poiseuille.m:248
    c.steps_left = max_steps > 0 ? max_steps : -1;
    if (!(c.vmax == c.vmax) || isinf(c.vmax)) c.status = SPHX_ERR_DIVERGED;
    c.dt = next_dt(c, ph);
    const int go = loop_continues(c) ? 1 : 0;
    c.run[0] = q0 == 0 ? go : 0;  // constant indices: a dynamically indexed member forces the struct into scratch
    c.run[1] = q0 == 0 ? 0 : go;
    c.pos_q[0] = (short)c.pos_count;     // (the first slot of the batch reads its parity's copy; the other one is rewritten before use)
    c.pos_q[1] = (short)c.pos_count;
    c.seq += 1;
    *clk = c;
    if (!go && c.pub) *c.pub = c;
}

// ---------------------------------------------------------------------------------------------
// The neighbour list of a step.  Pass A sweeps the candidates (3 cell columns of fluid, and of wall
// particles next to a wall) and packs the accepted ones into ONE list per particle: entry m belongs to
// lane m % LPP, row m / LPP, so every lane owns ceil-or-floor(cnt / LPP) neighbours in the later passes
// (balanced trip counts) and the lanes of a group walk adjacent particles.  Wall neighbours carry
// kWallBit: passes B..E walk one list and never touch the cell structure again (at a few thousand
// particles a pass is a chain of dependent memory round trips; cell -> row flag -> cell range -> wall
// particle was three of them).
// ---------------------------------------------------------------------------------------------
constexpr int kWallBit = 1 << 30;
// nl_cnt[lane]: rows this lane owns (low half) and how many of them, the first ones, hold fluid neighbours (high half)
__device__ __forceinline__ int list_rows(int packed) { return packed & 0x3fff; }
__device__ __forceinline__ int list_fluid_rows(int packed) { return packed >> 16; }
// Bits 14 and 15 of a lane's packed counts (slot-coded lists only, see kSlotCodes): some fluid entry of the lane's column of the
// SUPERSET list -- so of every list walked until the next re-binning -- names something beyond the first kForceSlots /
// kSlotCodes slots of the workgroup's layout.  Set by the cell sweep, handed on by pass A's walk.  A wavefront none of whose
// lanes has the bit reads its tile without looking at the entries (the `near` walks): the per-lane choice between the tile
// and global memory is a pair of exec-mask regions around every fetch and, because both arms load into the same registers,
// a wait for every outstanding global load -- the list words requested ahead -- in front of every LDS read.  Measured at 6 M
// particles with a timing-only build whose fetches never look (profiles/r04_pretend_lds_only_fetch_*.txt): pass A's walk
// 420 -> 369 us, KGC 252 -> 239, forces 522 -> 507, continuity 288 -> 275.
constexpr int kFarForcesBit = 1 << 14, kFarTileBit = 1 << 15;

// 16-bit entries of the large-channel lists (FluidTmp::nl_pk / sl_pk)
constexpr int kDeltaMax = 32767;
__device__ __forceinline__ void put_delta(int *pk, int stride, int row, int col, int d)
{
    reinterpret_cast<short *>(pk)[((size_t)((unsigned)row >> 1) * (unsigned)stride + (unsigned)col) * 2 + ((unsigned)row & 1u)] = (short)d;
}
__device__ __forceinline__ int delta_lo(int word) { return (int)(short)(word & 0xffff); }
__device__ __forceinline__ int delta_hi(int word) { return word >> 16; }
__device__ __forceinline__ int delta_of_row(const int *pk, int stride, int row, int col)
{
    const int word = pk[(size_t)(row >> 1) * stride + col];
    return (row & 1) ? delta_hi(word) : delta_lo(word);
}
// index difference k - i as stored (periodic grids: the representative nearest to zero modulo the population n)
__device__ __forceinline__ int encode_delta(int k, int i, int n, bool periodic, int *flags, int limit = kDeltaMax)
{
    int d = k - i;
    if (periodic) {
        const int half = n >> 1;
        if (d > half) d -= n;
        else if (d < -half) d += n;
    }
    if (d > limit || d < -limit) atomicOr(flags, 1);  // (ctx_setup keeps such grids on the compact kernels)
    return d;
}
// Slot-coded entries (template parameter CODED of the large-channel kernels; contexts whose four passes all stage LDS tiles,
// sphx_ctx::coded_lists).  The passes of such a context spent 13 vector instructions per neighbour turning its index into a
// tile slot (TileMap::slot: three ranges), four times a step -- and the answer does not change between two re-binnings.
// So the cell sweep, which runs once per re-binning, stores the answer: a 16-bit entry below kSlotCodes IS the neighbour's
// position in the workgroup's tile layout (tile_ranges at capacity kSlotCodes: own column, left, right -- a pass with a
// smaller tile stages a prefix of the same layout); any other value is the index difference + kCodeBias (neighbours the
// layout does not reach: a workgroup running across a column end, ranges longer than the layout).
#ifndef SPHX_EXP_PRETEND_COMPLETE_TILE
#define SPHX_EXP_PRETEND_COMPLETE_TILE 0  // measurement builds only (tools/probes/build_variant_lib.sh), see k_forces_w
#endif
constexpr int kSlotCodes = 480;
constexpr int kForceSlots = 464;  // the force pass stages this many (88-byte records: four workgroups of 40 864 B are the 160 KB of a CU)
constexpr int kCodeBias = 33000;
constexpr int kCodedDeltaMax = 32500;  // kSlotCodes <= kCodeBias - kCodedDeltaMax,  kCodeBias + kCodedDeltaMax <= 65535
__device__ __forceinline__ int code_lo(int word) { return word & 0xffff; }
__device__ __forceinline__ int code_hi(int word) { return (int)((unsigned)word >> 16); }
__device__ __forceinline__ int code_of_row(const int *pk, int stride, int row, int col)
{
    const int word = pk[(size_t)(row >> 1) * stride + col];
    return (row & 1) ? code_hi(word) : code_lo(word);
}
// ... and back: only wavefronts near the periodic seam can hold a particle whose neighbours wrap (see near_seam)
__device__ __forceinline__ int wrap_index(int k, int n) { return k < 0 ? k + n : (k >= n ? k - n : k); }

// Prologue shared by the passes: everything whose address does not depend on the clock is requested
// BEFORE the run flag is looked at, so the clock read overlaps the particle's own loads.
#define SPHX_PASS_INDEX_AT(bid, nblk)                                      \
    const int blk = xcd_block((bid), (nblk));                              \
    const int tid = blk * kBlock + threadIdx.x;                            \
    const int i = tid / LPP, sub = tid % LPP;                              \
    const bool in_cap = i < t.cap
#define SPHX_PASS_INDEX() SPHX_PASS_INDEX_AT((int)blockIdx.x, (int)gridDim.x)

// ---------------------------------------------------------------------------------------------
// pass A: candidate sweep -> neighbour list; number-density summation -> rho, Vol
// (mex/sph_physics_mex.c:188-234) and the half-step density/pressure of integration_1st's pre-pass
// (:857-862), which only needs own-particle data.
// ---------------------------------------------------------------------------------------------
// MODE 0: sweep the cells, write the step's list.  MODE 1: same sweep, also write the superset list.
// MODE 2: walk the superset list instead of the cells.
// bid / nblk: the workgroup's index among the nblk workgroups of this pass (the fused launch k_continuity_density runs the
// pass in a sub-range of its grid).  half: also write the half-step density / pressure (needs the step's dt; the fused
// launch runs pass A of the NEXT step, whose dt is not known yet -- pass B completes the record there).
template <int LPP, int MODE>
__device__ __forceinline__ void density_body(const Clock *clk, int q, const Grid &g, const Phys &ph, const FluidSet &s,
                                             const FluidTmp &t, const Walls &w, int bid, int nblk, bool half)
{

    SPHX_PASS_INDEX_AT(bid, nblk);
    const double2 pi = in_cap ? s.pos[i] : make_double2(0.0, 0.0);
    const int ci = in_cap ? s.cell[i] : 0;
    const bool lead = in_cap && sub == 0;  // the lane that finishes the particle
    const double mass_i = lead ? s.mass[i] : 1.0, drho_i = lead ? s.drho[i] : 0.0;
    // list-walking variant: row count and the first two rows are requested here as well (a lane rarely owns more
    // at 32 lanes per particle): count -> entry -> position becomes {count, entries} -> position
    int ns = 0, e_row0 = 0, e_row1 = 0;
    if (MODE == 2) {
        ns = list_rows(t.sl_cnt[tid]);
        e_row0 = t.sl_idx[tid];
        e_row1 = t.sl_idx[(size_t)t.nl_stride + tid];
    }
    const double dt = clk->dt;
    if (!clk->run[q]) return;
    const bool active = i < clk->n;
    double s_in = 0.0, s_ct = 0.0;
    int cnt = 0, scnt = 0, cnt_fl = 0, scnt_fl = 0;
    const int lane = threadIdx.x & 63, gbase = lane & ~(LPP - 1);
    const int row_base = tid - sub;  // list column of lane 0 of this group
    // the LPP lanes of a particle test LPP consecutive candidates at a time (uniform trip count over the
    // group); accepted ones are packed with ballot + popcount
    // a group (LPP <= 32 aligned lanes) lies in one 32-bit half of the wave's ballot: 32-bit shifts / popcounts
    // (the candidate loop of the sweeping variants is VALU-bound)
    const int half_shift = gbase & 31;
    const unsigned grp_mask = LPP >= 32 ? 0xffffffffu : ((1u << (LPP & 31)) - 1u);
    const unsigned below_me = (1u << sub) - 1u;  // sub <= 31
    auto group_bits = [&](bool acc) -> unsigned {
        const unsigned long long bal = __ballot(acc);
        const unsigned half = lane < 32 ? (unsigned)bal : (unsigned)(bal >> 32);
        return (half >> half_shift) & grp_mask;
    };
    auto push = [&](bool acc, int entry) {
        if (LPP == 1) {
            if (acc) {
                if (cnt < t.nl_cap) t.nl_idx[(size_t)cnt * t.nl_stride + tid] = entry;
                ++cnt;
            }
        } else {
            const unsigned grp = group_bits(acc);
            if (acc) {
                const int m = cnt + __popc(grp & below_me);
                if (m / LPP < t.nl_cap) t.nl_idx[(size_t)(m / LPP) * t.nl_stride + row_base + (m % LPP)] = entry;
            }
            cnt += __popc(grp);
        }
    };
    auto push_super = [&](bool acc, int entry) {
        if (LPP == 1) {
            if (acc) {
                if (scnt < t.sl_cap) t.sl_idx[(size_t)scnt * t.nl_stride + tid] = entry;
                ++scnt;
            }
        } else {
            const unsigned grp = group_bits(acc);
            if (acc) {
                const int m = scnt + __popc(grp & below_me);
                if (m / LPP < t.sl_cap) t.sl_idx[(size_t)(m / LPP) * t.nl_stride + row_base + (m % LPP)] = entry;
            }
            scnt += __popc(grp);
        }
    };
    constexpr bool walk = MODE == 2;
    constexpr bool record = MODE == 1;  // also write the superset list
    if (walk) {
        const int rows = group_extreme<LPP, true>(ns);  // (= lane 0's: it owns the most rows)
        if (active) {
            const double xi = pi.x, yi = pi.y;
            for (int m = 0; m < rows; ++m) {
                bool acc = false;
                int e = 0;
                if (m < ns) {
                    e = m == 0 ? e_row0 : (m == 1 ? e_row1 : t.sl_idx[(size_t)m * t.nl_stride + tid]);
                    const bool wall = (e & kWallBit) != 0;
                    const int k = e & (kWallBit - 1);
                    const double2 pj = (wall ? w.pos : (const double2 *)s.pos)[k];
                    const double Volw = wall ? w.a[k].x : 0.0;  // requested with the position, not after the distance test
                    const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
                    const double r2 = dx * dx + dy * dy;
                    if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                        acc = true;
                        const double W = spline_W_sel(ph.kc, r2 * rsqrt_nr(r2));
                        if (wall) s_ct += W * Volw;
                        else s_in += W;
                    }
                }
                // wall candidates sit behind the fluid ones: the fluid count is the count reached before the first row
                // that holds a wall entry, plus that row's accepted fluid entries
                const bool wall_row = (e & kWallBit) != 0;
                if (!__any(wall_row)) {
                    push(acc, e);
                    cnt_fl = cnt;
                } else {
                    cnt_fl += LPP == 1 ? ((acc && !wall_row) ? 1 : 0) : __popc(group_bits(acc && !wall_row));
                    push(acc, e);
                }
            }
            if (cnt > t.nl_cap * LPP) { atomicOr(t.flags, 1); cnt = t.nl_cap * LPP; }
            cnt_fl = min(cnt_fl, cnt);
        }
    } else if (active) {
        const double xi = pi.x, yi = pi.y;
        const int cx = ci / g.ncy, cy = ci - cx * g.ncy;
        const int cylo = max(cy - 1, 0), cyhi = min(cy + 1, g.ncy - 1);
        const bool near_wall = w.row_any[cy] != 0;
        // cell ranges of the three columns, fluid and wall, requested together
        int lo[3], hi[3], wlo[3], whi[3];
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox) {
            int col = cx + ox;
            bool ok = true;
            if (g.periodic) {
                if (duplicate_column(g, ox)) ok = false;
                if (col < 0) col += g.ncx;
                else if (col >= g.ncx) col -= g.ncx;
            } else if (col < 0 || col >= g.ncx) {
                ok = false;
            }
            const int c0 = col * g.ncy + cylo, c1 = col * g.ncy + cyhi + 1;
            lo[ox + 1] = ok ? s.start[c0] : 0;
            hi[ox + 1] = ok ? s.start[c1] : 0;
            wlo[ox + 1] = ok ? w.start[c0] : 0;
            whi[ox + 1] = ok ? w.start[c1] : 0;
        }
        // The three fluid ranges (and, next to a wall, the three wall ranges) are walked as ONE virtual index range:
        // a group of 32 lanes then needs one or two trips instead of one or two per column, and each trip's
        // loads are independent of the previous trip's packing.
        const int n0 = hi[0] - lo[0], n1 = hi[1] - lo[1], n2 = hi[2] - lo[2];
        const int nfl = n0 + n1 + n2;
        for (int vb = 0; vb < nfl; vb += LPP) {
            const int v = vb + sub;
            bool acc = false, wide = false;
            int k = 0;
            if (v < nfl) {
                k = v < n0 ? lo[0] + v : (v < n0 + n1 ? lo[1] + (v - n0) : lo[2] + (v - n0 - n1));
                const double2 pj = s.pos[k];
                const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
                const double r2 = dx * dx + dy * dy;
                if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                    acc = true;
                    s_in += spline_W_sel(ph.kc, r2 * rsqrt_nr(r2));
                }
                wide = r2 > kR2Min && r2 < t.sl_rcut2;
            }
            push(acc, k);
            if (record) push_super(wide, k);
        }
        cnt_fl = cnt;
        scnt_fl = scnt;
        if (near_wall) {
            const int w0 = whi[0] - wlo[0], w1 = whi[1] - wlo[1], w2 = whi[2] - wlo[2];
            const int nwl = w0 + w1 + w2;
            for (int vb = 0; vb < nwl; vb += LPP) {
                const int v = vb + sub;
                bool acc = false, wide = false;
                int k = 0;
                if (v < nwl) {
                    k = v < w0 ? wlo[0] + v : (v < w0 + w1 ? wlo[1] + (v - w0) : wlo[2] + (v - w0 - w1));
                    const double2 pj = w.pos[k];
                    const double Volj = w.a[k].x;  // requested with the position, not after the distance test
                    const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
                    const double r2 = dx * dx + dy * dy;
                    if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                        acc = true;
                        s_ct += spline_W_sel(ph.kc, r2 * rsqrt_nr(r2)) * Volj;
                    }
                    wide = r2 > kR2Min && r2 < t.sl_rcut2;
                }
                push(acc, k | kWallBit);
                if (record) push_super(wide, k | kWallBit);
            }
        }
        if (cnt > t.nl_cap * LPP) { atomicOr(t.flags, 1); cnt = t.nl_cap * LPP; }
        cnt_fl = min(cnt_fl, cnt);
        if (record && scnt > t.sl_cap * LPP) { atomicOr(t.flags, 1); scnt = t.sl_cap * LPP; }
        scnt_fl = min(scnt_fl, scnt);
    }
    // cnt is the particle's neighbour count (identical in all lanes of the group); lane `sub` owns entries
    // sub, sub+LPP, ...  Fluid neighbours come first (cnt_fl of them), so a lane's rows are its fluid rows followed by
    // its wall rows: both counts are stored (list_rows / list_fluid_rows).
    if (tid < t.nl_stride)
        t.nl_cnt[tid] = (cnt > sub ? (cnt - sub + LPP - 1) / LPP : 0) | ((cnt_fl > sub ? (cnt_fl - sub + LPP - 1) / LPP : 0) << 16);
    if (record && tid < t.nl_stride)
        t.sl_cnt[tid] = (scnt > sub ? (scnt - sub + LPP - 1) / LPP : 0) | ((scnt_fl > sub ? (scnt_fl - sub + LPP - 1) / LPP : 0) << 16);
    s_in = group_sum<LPP>(s_in);
    s_ct = group_sum<LPP>(s_ct);
    if (active && sub == 0) {
        const double m = mass_i;
        const double rho = density_from_sigma(ph.w0 + s_in, s_ct, m, ph.rho0, ph.inv_sigma0);
        double rhoh = rho + 0.5 * dt * drho_i;
        if (rhoh < 1e-10) rhoh = ph.rho0;
        t.a[i] = half ? make_double4(m / rho, eos_pressure(rhoh, ph.rho0, ph.p0), rhoh, rho) : make_double4(m / rho, 0.0, 0.0, rho);
        t.vol[i] = m / rho;
    }
}

// the half-step density / pressure of a particle (integration_1st's pre-pass, sph_physics_mex.c:857-862)
__device__ __forceinline__ void half_state(const Phys &ph, double rho, double drho, double dt, double &rhoh, double &ph_)
{
    rhoh = rho + 0.5 * dt * drho;
    if (rhoh < 1e-10) rhoh = ph.rho0;
    ph_ = eos_pressure(rhoh, ph.rho0, ph.p0);
}

// Slabs, frozen steps: pass A of the next step in two launches, so that the halo exchange of this step hides behind the
// first.  part = 1: only the INTERIOR workgroups -- every particle of theirs binned in the columns [own_c0 + 1, own_c1 - 1),
// so that their candidates (one column either side) and the tiles they stage lie in owned columns, which the incoming message
// does not touch -- launched right behind k_slab_pack3; part = 2: the rest (boundary workgroups), behind k_slab_unpack3.
// part rides in the walk kernels' cond_fresh argument: bits 4 and up (kPassPartShift).
constexpr int kPassPartShift = 4;
template <int LPP>
__device__ __forceinline__ bool slab_part_skips(const Grid &g, const FluidSet &s, int blk, int part)
{
    const int first = blk * (kBlock / LPP), last = first + kBlock / LPP - 1;
    const int lo = s.start[(g.own_c0 + 1) * g.ncy], hi = s.start[(g.own_c1 - 1) * g.ncy];
    const bool interior = first >= lo && last < hi;
    return interior != (part == 1);
}

// cond_fresh (dynamic contexts launch both MODE 1 and MODE 2 on every step): -1 = always run, 1 = only on a fresh grid,
// 0 = only on a grid that is not fresh.  Measured at 6 M particles, average pass A per step: two launches of which one
// returns 786 us; one kernel holding both bodies 891 us (it runs at the register budget of the bigger one); one
// body deciding per candidate at run time 795 us.
template <int LPP, int MODE>
__global__ __launch_bounds__(kBlock) void k_density(const Clock *clk, int q, Grid g, Phys ph,
                                                    FluidSet s, FluidTmp t, Walls w, int cond_fresh)
{
    const int part = cond_fresh >= 0 ? cond_fresh >> kPassPartShift : 0;
    if (part) cond_fresh &= (1 << kPassPartShift) - 1;
    if (cond_fresh >= 0 && (clk->fresh != 0) != (cond_fresh != 0)) return;
    if (MODE == 2 && part && slab_part_skips<LPP>(g, s, xcd_block((int)blockIdx.x, (int)gridDim.x), part)) return;
    density_body<LPP, MODE>(clk, q, g, ph, s, t, w, (int)blockIdx.x, (int)gridDim.x, true);
}

// ---------------------------------------------------------------------------------------------
// pass B: kernel-gradient-correction matrix A -> blended pseudo-inverse B
// (mex/sph_physics_mex.c:239-366).  Fluid and wall neighbours contribute the same term (Vol_j of a wall
// particle is m/rho0), so the walk does not branch.
// ---------------------------------------------------------------------------------------------
// finish_half: pass A of this step ran inside the previous step's fused launch and left {p_half, rho_half} open
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_kgc(const Clock *clk, int q, Grid g, Phys ph, FluidSet s,
                                                FluidTmp t, Walls w, int finish_half)
{
    SPHX_PASS_INDEX();
    const double2 pi = in_cap ? s.pos[i] : make_double2(0.0, 0.0);
    const bool closes = finish_half && in_cap && sub == 0;
    const double4 a_own = closes ? t.a[i] : make_double4(0.0, 0.0, 0.0, 0.0);
    const double drho_own = closes ? s.drho[i] : 0.0;
    const int nn_all = list_rows(t.nl_cnt[tid]);
    // the first rows are requested together with the count (at 32 lanes per particle a lane rarely owns more than
    // two): count -> entry -> neighbour data becomes {count, entries} -> neighbour data
    const int e_row0 = t.nl_idx[tid], e_row1 = t.nl_idx[(size_t)t.nl_stride + tid];
    // (rows 2 and 3 only where lanes own that many: few lanes per particle)
    const int e_row2 = LPP <= 8 ? t.nl_idx[2 * (size_t)t.nl_stride + tid] : 0;
    const int e_row3 = LPP <= 8 ? t.nl_idx[3 * (size_t)t.nl_stride + tid] : 0;
    if (!clk->run[q]) return;
    const bool active = i < clk->n;
    double a11 = 0.0, a12 = 0.0, a21 = 0.0, a22 = 0.0;
    if (active) {
        const double xi = pi.x, yi = pi.y;
        for (int m = 0; m < nn_all; ++m) {
            const int e = m == 0 ? e_row0 : (m == 1 ? e_row1 : (LPP <= 8 && m == 2 ? e_row2 : (LPP <= 8 && m == 3 ? e_row3 : t.nl_idx[(size_t)m * t.nl_stride + tid])));
            const bool wall = (e & kWallBit) != 0;
            const int k = e & (kWallBit - 1);
            const double2 pj = (wall ? w.pos : (const double2 *)s.pos)[k];
            const double Volj = wall ? w.a[k].x : t.vol[k];
            const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
            const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
            const double ex = dx * inv_r, ey = dy * inv_r;
            const double fxj = spline_dW_in(ph.kc, r) * Volj;
            a11 -= dx * (fxj * ex);
            a12 -= dx * (fxj * ey);
            a21 -= dy * (fxj * ex);
            a22 -= dy * (fxj * ey);
        }
    }
    a11 = group_sum<LPP>(a11);
    a12 = group_sum<LPP>(a12);
    a21 = group_sum<LPP>(a21);
    a22 = group_sum<LPP>(a22);
    if (active && sub == 0) {
        const Mat2 B = kgc_from_A(a11, a12, a21, a22);
        t.B[i] = make_double4(B.m11, B.m12, B.m21, B.m22);
        if (finish_half) {
            double rhoh, p_half;
            half_state(ph, a_own.w, drho_own, clk->dt, rhoh, p_half);
            t.a[i] = make_double4(a_own.x, p_half, rhoh, a_own.w);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pass CD: viscous force (+gravity) [sph_physics_mex.c:469-545, SPH_Poiseuille.m:392], transport
// shift [:636-710], Riemann pressure force of integration_1st [:870-957], velocity kick
// [:1400-1408] and both position half-drifts [:863-864,:1066-1069] + periodic wrap
// [SPH_Poiseuille.m:570-577].  One walk over the list serves all three operators because they share
// e, dW, B_i+B_j; the wall entries are visited a second time because the wall pressure needs the
// complete viscous+gravity force of the particle first (p_wall uses force_prior_i, :931-934).
// ---------------------------------------------------------------------------------------------
// later = 1 (dual-rate loop, inner sub-steps after the first): the pressure part only -- viscous force and gravity
// are those of the first sub-step (t.fp), there is no transport shift, the particle moves on from t.posn; pair geometry
// stays that of the start of the outer step (s.pos), velocities are the latest ones (s.vel = the previous sub-step's).
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_forces(const Clock *clk, int q, Grid g, Phys ph, FluidSet s,
                                                   FluidTmp t, Walls w, int later)
{
    SPHX_PASS_INDEX();
    const double2 pi = in_cap ? s.pos[i] : make_double2(0.0, 0.0);
    const double2 vi = in_cap ? s.vel[i] : make_double2(0.0, 0.0);
    const double4 ai = in_cap ? t.a[i] : make_double4(1.0, 0.0, 0.0, 0.0);
    const double4 Bi = in_cap ? t.B[i] : make_double4(1.0, 0.0, 0.0, 1.0);
    const double mi = in_cap ? s.mass[i] : 1.0;
    const int nn_all = list_rows(t.nl_cnt[tid]);
    // the first rows are requested together with the count (at 32 lanes per particle a lane rarely owns more than
    // two): count -> entry -> neighbour data becomes {count, entries} -> neighbour data
    const int e_row0 = t.nl_idx[tid], e_row1 = t.nl_idx[(size_t)t.nl_stride + tid];
    // (rows 2 and 3 only where lanes own that many: few lanes per particle)
    const int e_row2 = LPP <= 8 ? t.nl_idx[2 * (size_t)t.nl_stride + tid] : 0;
    const int e_row3 = LPP <= 8 ? t.nl_idx[3 * (size_t)t.nl_stride + tid] : 0;
    const bool tracked = s.posb != nullptr;
    const double2 pb = (tracked && in_cap && sub == 0) ? s.posb[i] : make_double2(0.0, 0.0);
    const double2 fp_own = (later && in_cap) ? t.fp[i] : make_double2(0.0, 0.0);
    const double2 p_now = (later && in_cap && sub == 0) ? t.posn[i] : make_double2(0.0, 0.0);
    const double dt = clk->dt;
    if (!clk->run[q]) return;
    const bool active = i < clk->n;
    const double h = ph.kc.h;
    double ax = 0.0, ay = 0.0, ix = 0.0, iy = 0.0, px = 0.0, py = 0.0, d2 = 0.0;
    const double xi = pi.x, yi = pi.y, vxi = vi.x, vyi = vi.y;
    const double Voli = ai.x, p_i = ai.y, rhoh_i = ai.z;
    const double b11i = Bi.x, b12i = Bi.y, b21i = Bi.z, b22i = Bi.w;
    int first_wall = 1 << 20;  // wall neighbours are appended behind the fluid ones: this lane's rows >= first_wall
    if (active) {
        for (int m = 0; m < nn_all; ++m) {
            const int e = m == 0 ? e_row0 : (m == 1 ? e_row1 : (LPP <= 8 && m == 2 ? e_row2 : (LPP <= 8 && m == 3 ? e_row3 : t.nl_idx[(size_t)m * t.nl_stride + tid])));
            const int k = e & (kWallBit - 1);
            if (!(e & kWallBit)) {
                const double2 pj = s.pos[k], vj = s.vel[k];
                const double4 aj = t.a[k], Bj = t.B[k];
                const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
                const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
                const double ex = dx * inv_r, ey = dy * inv_r;
                const double dW = spline_dW_in(ph.kc, r);
                const double Volj = aj.x;
                const double tx = (b11i + Bj.x) * ex + (b12i + Bj.y) * ey;
                const double ty = (b21i + Bj.z) * ex + (b22i + Bj.w) * ey;
                const double eBe = ex * tx + ey * ty;
                const double vxj = vj.x, vyj = vj.y;
                const double dWVj = dW * Volj;
                if (!later) {
                    // viscous
                    const double coeff = eBe * ph.mu * dWVj * rcp_nr(r + 0.01 * h);
                    ax += coeff * (vxi - vxj);
                    ay += coeff * (vyi - vyj);
                    // transport
                    ix -= dWVj * tx;
                    iy -= dWVj * ty;
                }
                // pressure (Riemann-dissipated face pressure)
                const double p_j = aj.y;
                const double rho_bar = 0.5 * (rhoh_i + aj.z);
                const double un_l = vxi * ex + vyi * ey, un_r = vxj * ex + vyj * ey;
                const double beta = riemann_beta(un_l, un_r, ph.c_f);
                const double p_avg = 0.5 * (p_i + p_j);
                const double p_star = p_avg + 0.5 * beta * rho_bar * (un_l - un_r);
                const double p_face = 0.5 * (p_avg + p_star);
                px -= (p_face * tx) * dWVj;
                py -= (p_face * ty) * dWVj;
            } else {
                first_wall = min(first_wall, m);
                if (later) continue;  // (wall entries contribute to the pressure part in the second loop only)
                const double2 pj = w.pos[k];
                const double4 wj = w.a[k];
                const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
                const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
                const double ex = dx * inv_r, ey = dy * inv_r;
                const double dWVj = spline_dW_in(ph.kc, r) * wj.x;
                const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
                const double eBe = ex * tx + ey * ty;
                const double coeff = 4.0 * eBe * ph.mu * dWVj * rcp_nr(r + 0.01 * h);
                ax += coeff * (vxi - wj.y);
                ay += coeff * (vyi - wj.z);
                ix -= 2.0 * dWVj * tx;
                iy -= 2.0 * dWVj * ty;
            }
        }
    }
    ax = group_sum<LPP>(ax);
    ay = group_sum<LPP>(ay);
    ix = group_sum<LPP>(ix);
    iy = group_sum<LPP>(iy);
    const double fpx = later ? fp_own.x : ax * Voli + mi * ph.g;  // + gravity, SPH_Poiseuille.m:392
    const double fpy = later ? fp_own.y : ay * Voli;
    const double inv_m = rcp_nr(mi);
    if (active) {
        const double acx = fpx * inv_m, acy = fpy * inv_m;
        for (int m = first_wall; m < nn_all; ++m) {
            const int k = t.nl_idx[(size_t)m * t.nl_stride + tid] & (kWallBit - 1);
            const double2 pj = w.pos[k];
            const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
            const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
            const double ex = dx * inv_r, ey = dy * inv_r;
            const double dWVj = spline_dW_in(ph.kc, r) * w.a[k].x;
            const double face = -(acx * ex + acy * ey);
            const double p_wall = p_i + rhoh_i * r * fmax(0.0, face);
            const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
            px -= (p_i + p_wall) * dWVj * tx;
            py -= (p_i + p_wall) * dWVj * ty;
        }
    }
    px = group_sum<LPP>(px);
    py = group_sum<LPP>(py);
    if (active && sub == 0) {
        const double fx = px * Voli, fy = py * Voli;
        const double vxn = vxi + (fpx + fx) * inv_m * dt;
        const double vyn = vyi + (fpy + fy) * inv_m * dt;
        double sx = 0.0, sy = 0.0;
        if (!later) transport_shift(ix, iy, h, ph.tc, sx, sy);
        double xo = later ? p_now.x : xi + sx, yo = later ? p_now.y : yi + sy;
        xo += 0.5 * dt * vxi;
        yo += 0.5 * dt * vyi;
        xo += 0.5 * dt * vxn;
        yo += 0.5 * dt * vyn;
        if (tracked && (!g.own_by_cell || owns(g, 0.0, s.cell[i]))) {  // (a slab bounds the drift of what it owns)
            const double ddx = min_image(g, xo - pb.x), ddy = yo - pb.y;
            d2 = ddx * ddx + ddy * ddy;
            if (d2 != d2) d2 = INFINITY;
        }
        t.posn[i] = make_double2(g.periodic ? wrap_x(xo, ph.DL) : xo, yo);  // a slab wraps when particles change owner
        t.veln[i] = make_double2(vxn, vyn);
        if (!later) t.fp[i] = make_double2(fpx, fpy);
        t.f[i] = make_double2(fx, fy);
    }
    // largest drift from the binning positions (bounds how stale the cell grid may get, see Clock::drift)
    d2 = wave_max(d2);
    __shared__ double s_d2[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s_d2[threadIdx.x >> 6] = d2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_d2[0];
        for (int k = 1; k < kBlock / 64; ++k) m = fmax(m, s_d2[k]);
        t.dpart[blk] = m;
    }
}

// =================================================================================================
// Large channels (few lanes per particle, LPP <= 8): the "_w" variants of passes B, CD and E.
// Counters of the round-1 kernels at 0.5-6 M particles: VALU 55-75 % busy, texture addresser 60-90 %, and a wave
// spending ~5 x its own issue time in flight -- every trip of the walk was two dependent memory round trips
// (list entry -> neighbour records) followed by 55-100 VALU instructions under per-entry fluid/wall and kernel-piece
// branches.  Here
//   * list entries run three rows ahead of their use (one register each): one round trip per trip;
//   * fluid and wall neighbours have loops of their own (fluid entries come first in a lane's rows, pass A stores both
//     row counts) -- no per-entry branch, no pointer selects;
//   * the minimum-image fold runs only in wavefronts holding a particle near the periodic seam;
//   * the kernel derivative is branch-free, 1/sqrt and the viscous division are Newton-refined hardware estimates;
//   * the force pass can stage its tile's neighbourhood in LDS (TILE > 0), see k_forces_w.
// Small channels keep the compact kernels above: at a few thousand particles a pass is a chain of dependent round
// trips and code size (instruction-cache fill per launch) matters more than instruction count.
// =================================================================================================

// Large-channel kernels are launched for the CAPACITY of the arrays; a workgroup whose first particle lies beyond the
// current population has nothing to do.  Its lanes used to request their own records like everybody else before looking
// at the clock (the latency trick of the small channels) -- on a slab, whose capacity holds 15-20 % of slack, that was
// 15-20 % more waves going through every pass's prologue.
// (only where there is slack -- FluidTmp::has_slack: the check is a look at the clock before anything else is requested)
template <int LPP>
__device__ __forceinline__ bool beyond_population(const Clock *clk, const FluidTmp &t, int blk)
{
    return t.has_slack && blk * (kBlock / LPP) >= clk->n;
}

// A neighbour's x can be a period away from the particle's only if one of them was binned in the first or last cell
// column and has since been wrapped; particles whose x lies within 2.5 columns of either end of the period cover every
// such pair (a column is wider than the skin, so a particle binned in column <= 1 is still left of 2.5 columns).
__device__ __forceinline__ bool near_seam(const Grid &g, double x)
{
    const double w = 2.5 / g.inv_csx;
    return g.periodic && (x < w || x > g.DL - w);
}

// Rows [0, rows_fl) of this lane's list column hold fluid neighbours as index differences, two rows per word (nl_pk):
// body(i + d) for each, the words two ahead of their use (= rows 4-5 ahead).  w0 / w1: the first two words, requested by
// the caller together with the row counts.
// (CODED: body(code), see kSlotCodes)
template <bool CODED = false, typename Body>
__device__ __forceinline__ void walk_fluid_rows(const FluidTmp &t, int tid, int i, int rows_fl, int w0, int w1, Body &&body)
{
    if (rows_fl <= 0) return;
    const int n_words = (rows_fl + 1) >> 1;
    int wa = w0, wb = w1;
    for (int p = 0; p < n_words; ++p) {
        const int wc = p + 2 < n_words ? t.nl_pk[(size_t)(p + 2) * t.nl_stride + tid] : 0;
        body(CODED ? code_lo(wa) : i + delta_lo(wa));
        if (2 * p + 1 < rows_fl) body(CODED ? code_hi(wa) : i + delta_hi(wa));
        wa = wb; wb = wc;
    }
}

// Rows [rows_fl, rows) hold wall neighbours (only next to a wall): body(k)
template <typename Body>
__device__ __forceinline__ void walk_wall_rows(const FluidTmp &t, int tid, int rows_fl, int rows, Body &&body)
{
    for (int m = rows_fl; m < rows; ++m) body(t.nl_idx[(size_t)m * t.nl_stride + tid] & (kWallBit - 1));
}

// LDS tile of a neighbour pass.  The workgroup's particles are consecutive in cell order -- a run of cells of one
// column -- and every fluid neighbour (or superset candidate) of theirs was binned into the same rows (+-1) of the three
// columns around it: three CONTIGUOUS index ranges.  The first `cap` particles of those ranges (own column first) are
// copied into LDS with coalesced loads once per workgroup and the walk gathers from there; a neighbour outside the staged
// ranges (a tile running across a column end, ranges longer than the tile) is read from global memory, so results do
// not depend on the tile size, bit for bit.  Measured (us per launch, with / without the tile):
//   6 M particles, 2 lanes per particle: forces 607 / 872, KGC 295 / 320, continuity 375 / 452, pass A walk 475 / 434
//   6 M particles, 4 lanes            : forces 673 / 740, KGC 372 / 344, continuity 465 / 446, pass A walk 511 / 451
//   0.5 M particles, 2 lanes          : forces 57.1 / 64.2, KGC 33.1 / 29.0, continuity 36.7 / 37.0, walk 51.8 / 46.6
// -- the fewer lanes share a particle the less its global gathers coalesce, and only records of >= 40 bytes per neighbour
// repay the staging: the force pass always uses the tile, KGC and continuity at 2 lanes per particle on channels that do
// not fit the Infinity Cache, pass A never.  (The 6 M figures without entries running ahead: 803 with / 784 without --
// the exposed round trip was the list entry, not the gather.)
struct TileMap {
    int lo0, lo1, lo2, len0, len1, len2;
    __device__ __forceinline__ int total() const { return len0 + len1 + len2; }
    __device__ __forceinline__ int index(int sl) const  // staged slot -> particle
    {
        return sl < len0 ? lo0 + sl : (sl < len0 + len1 ? lo1 + (sl - len0) : lo2 + (sl - len0 - len1));
    }
    __device__ __forceinline__ int slot(int k) const  // particle -> staged slot, -1: not staged
    {
        // three independent selects, not a nested choice: the nested form compiles to a chain of exec-mask branches (five
        // per neighbour in the walks, with their s_nop hazards), this one to compares and v_cndmask
        const unsigned u0 = (unsigned)(k - lo0), u1 = (unsigned)(k - lo1), u2 = (unsigned)(k - lo2);
        int sl = -1;
        sl = u2 < (unsigned)len2 ? len0 + len1 + (int)u2 : sl;
        sl = u1 < (unsigned)len1 ? len0 + (int)u1 : sl;
        sl = u0 < (unsigned)len0 ? (int)u0 : sl;
        return sl;
    }
    __device__ __forceinline__ TileMap capped(int cap) const  // the first `cap` slots of the same layout
    {
        TileMap m = *this;
        m.len0 = min(len0, cap);
        m.len1 = min(len1, cap - m.len0);
        m.len2 = min(len2, cap - m.len0 - m.len1);
        return m;
    }
};
#ifndef SPHX_EXP_N2_PROLOGUE
#define SPHX_EXP_N2_PROLOGUE 0  // measurement builds only (tools/probes/build_variant_lib.sh)
#endif
// MEASUREMENT ONLY: what decoding a layout of 3 + 14 ranges would add to every staged slot (a blocked particle order's tile
// neighbourhood: the strip's own run plus one cell above and below per column).  Fourteen compare / select /
// add steps on values the compiler cannot fold; the result is 0, so the staged data -- and every result -- stay what they are.
__device__ __forceinline__ int n2_prologue_cost(int sl, const TileMap &m)
{
#if SPHX_EXP_N2_PROLOGUE
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 14; ++j) acc = sl >= m.len0 + m.len1 + m.len2 + 1000 + 37 * j ? acc + m.lo0 + j : acc;
    return acc;
#else
    return 0;
#endif
}
// The layout (tile_ranges at capacity kSlotCodes) of every workgroup is worked out ONCE per re-binning, by the cell sweep, and
// kept in FluidTmp::tmap.  Working it out in every pass put two dependent memory round trips (first / last cell of the
// workgroup -> six cell starts) in front of the staging loads, in a prologue that already waits for the clock: the workgroups
// of a compute unit start together and stay in step, so while they all wait nobody computes (6 M particles: a workgroup of
// the force pass lives ~16 us, of which the vector units have work for ~9).  Read here with the workgroup's first requests.
__device__ __forceinline__ TileMap tile_map_of(const FluidTmp &t, int blk)
{
    const int4 lo = reinterpret_cast<const int4 *>(t.tmap)[2 * blk], len = reinterpret_cast<const int4 *>(t.tmap)[2 * blk + 1];
    return TileMap{lo.x, lo.y, lo.z, len.x, len.y, len.z};
}
__device__ __forceinline__ void store_tile_map(const FluidTmp &t, int blk, const TileMap &m)
{
    reinterpret_cast<int4 *>(t.tmap)[2 * blk] = make_int4(m.lo0, m.lo1, m.lo2, 0);
    reinterpret_cast<int4 *>(t.tmap)[2 * blk + 1] = make_int4(m.len0, m.len1, m.len2, 0);
}
// ... cut to a pass's tile size (nothing for a workgroup beyond the population)
template <int LPP>
__device__ __forceinline__ TileMap staged_map(const TileMap &layout, int blk, int n_now, int cap)
{
    if (blk * (kBlock / LPP) >= n_now) return TileMap{0, 0, 0, 0, 0, 0};
    return layout.capped(cap);
}

// the particle a slot-coded entry of particle i names (full: the layout at capacity kSlotCodes); the passes call this only for
// the entries their tile does not hold
__device__ __forceinline__ int coded_index(const TileMap &full, int code, int i, int n)
{
    // (slot -> particle as two independent selects on the layout's range ends, like fluid_index of the sweep: TileMap::index
    // is a nested choice and compiles to exec-mask branches, five of them per neighbour the force pass does not stage)
    const int n0 = full.len0, n01 = full.len0 + full.len1;
    const int o0 = full.lo0, o1 = full.lo1 - n0, o2 = full.lo2 - n01;
    const int k_slot = code + o2 + (code < n01 ? o1 - o2 : 0) + (code < n0 ? o0 - o1 : 0);
    const int k_far = wrap_index(i + code - kCodeBias, n);
    return code < kSlotCodes ? k_slot : k_far;
}

template <int LPP>
__device__ __forceinline__ TileMap tile_ranges(const Grid &g, const FluidSet &s, int blk, int n_now, int cap)
{
    TileMap tm{0, 0, 0, 0, 0, 0};
    constexpr int per_tile = kBlock / LPP;
    const int p_first = blk * per_tile;
    if (p_first >= n_now) return tm;
    const int p_last = min(p_first + per_tile, n_now) - 1;
    const int c_first = s.cell[p_first], c_last = s.cell[p_last];
    const int cx = c_first / g.ncy, r_lo = c_first - cx * g.ncy;
    const int r_hi = (c_last / g.ncy == cx) ? c_last - cx * g.ncy : g.ncy - 1;  // the tile's part in column cx
    const int ra = max(r_lo - 1, 0), rb = min(r_hi + 1, g.ncy - 1) + 1;
    int lo[3], len[3];
#pragma unroll
    for (int ox = -1; ox <= 1; ++ox) {
        int col = cx + ox;
        bool ok = true;
        if (g.periodic) {
            if (duplicate_column(g, ox)) ok = false;
            if (col < 0) col += g.ncx;
            else if (col >= g.ncx) col -= g.ncx;
        } else if (col < 0 || col >= g.ncx) {
            ok = false;
        }
        const int a0 = ok ? s.start[col * g.ncy + ra] : 0, a1 = ok ? s.start[col * g.ncy + rb] : 0;
        lo[ox + 1] = a0;
        len[ox + 1] = a1 - a0;
    }
    tm.lo0 = lo[1]; tm.len0 = min(len[1], cap);
    tm.lo1 = lo[0]; tm.len1 = min(len[0], cap - tm.len0);
    tm.lo2 = lo[2]; tm.len2 = min(len[2], cap - tm.len0 - tm.len1);
    return tm;
}

// Tile arrays that reach a device function as plain pointer arguments (continuity_body, density_walk_body: the same bodies
// serve kernels with and without a tile) are generic pointers to the compiler, and it read them with FLAT loads -- the texture
// path plus an aperture check -- instead of ds_read (round 3: k_continuity<2, true, 448> at 6 M particles 415 -> 376 us once
// told).  Read them through LDS-qualified pointers.
using lds_f64 = const __attribute__((address_space(3))) double;
__device__ __forceinline__ double2 lds_double2(const double2 *tile, int slot)
{
    lds_f64 *p = (lds_f64 *)tile;
    return make_double2(p[2 * slot], p[2 * slot + 1]);
}
__device__ __forceinline__ double lds_double(const double *tile, int slot) { return ((lds_f64 *)tile)[slot]; }
__device__ __forceinline__ double4 lds_double4(const double4 *tile, int slot)
{
    lds_f64 *p = (lds_f64 *)tile;
    return make_double4(p[4 * slot], p[4 * slot + 1], p[4 * slot + 2], p[4 * slot + 3]);
}

// tile capacity in particles: the three-column neighbourhood of kBlock / LPP particles at ~9 particles per cell
__host__ __device__ constexpr int tile_slots(int lpp) { return lpp >= 8 ? 192 : (lpp == 4 ? 320 : 480); }

// pass A, walking the superset list (see density_body MODE 2).  The LPP lanes of a particle test one list row at a
// time and pack the accepted entries with ballot + popcount, so a group's trip count is uniform: the rows in which all
// of its lanes hold fluid candidates run a branch-free fluid body with the entries three rows ahead, the (at most one
// mixed + the wall) rows behind them run the general body.
// (LDS tile: staging the candidate positions makes this pass slower at 0.5 M particles -- 50.5 against 42.3 us, it gathers
// only 16 bytes per candidate -- and, now that the pass runs at the texture addresser's limit, 6 % faster at 6 M: used
// where KGC and continuity use theirs.)
// (CODED: both lists hold slot-coded entries, see kSlotCodes -- the walk copies them as they stand)
template <int LPP, int TILE = 0, bool CODED = false>
__device__ __forceinline__ void density_walk_body(const Clock *clk, int q, const Grid &g, const Phys &ph, const FluidSet &s,
                                                  const FluidTmp &t, const Walls &w, int bid, int nblk, bool half,
                                                  double2 *c_pos = nullptr)
{
    static_assert(!CODED || TILE == kSlotCodes, "slot-coded lists: the walk stages the whole layout");
    SPHX_PASS_INDEX_AT(bid, nblk);
    if (beyond_population<LPP>(clk, t, blk)) return;  // (nl_cnt of lanes beyond the population is never looked at)
    const double2 pi = in_cap ? s.pos[i] : make_double2(0.0, 0.0);
    const bool lead = in_cap && sub == 0;
    const double mass_i = lead ? s.mass[i] : 1.0, drho_i = lead ? s.drho[i] : 0.0;
    const int spacked = t.sl_cnt[tid];
    const int w_first0 = t.sl_pk[tid], w_first1 = t.sl_pk[(size_t)t.nl_stride + tid];  // rows 0-3 of the packed fluid rows
    const TileMap layout = TILE > 0 ? tile_map_of(t, blk) : TileMap{0, 0, 0, 0, 0, 0};
    const double dt = clk->dt;
    if (!clk->run[q]) return;
    const int n_now = clk->n;
    const bool active = i < n_now;
    TileMap tm{0, 0, 0, 0, 0, 0};
    if (TILE > 0) {  // candidate positions of the workgroup's three-column neighbourhood staged in LDS (see tile_ranges)
        tm = staged_map<LPP>(layout, blk, n_now, TILE);
        static_assert(TILE <= 2 * kBlock, "two slots per thread");
        {  // (both of a thread's slots requested before either is waited for: see k_forces_w)
            const int total = tm.total(), sl0 = threadIdx.x, sl1 = threadIdx.x + kBlock;
            const bool h0 = sl0 < total, h1 = sl1 < total;
            const int k0 = tm.index(h0 ? sl0 : 0) + n2_prologue_cost(sl0, tm), k1 = tm.index(h1 ? sl1 : 0) + n2_prologue_cost(sl1, tm);
            double2 p0, p1;
            if (h0) p0 = s.pos[k0];
            if (h1) p1 = s.pos[k1];
            if (h0) c_pos[sl0] = p0;
            if (h1) c_pos[sl1] = p1;
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, gbase = lane & ~(LPP - 1);
    const int row_base = tid - sub;
    const int half_shift = gbase & 31;
    const unsigned grp_mask = LPP >= 32 ? 0xffffffffu : ((1u << (LPP & 31)) - 1u);
    const unsigned below_me = (1u << sub) - 1u;
    auto group_bits = [&](bool acc) -> unsigned {
        const unsigned long long bal = __ballot(acc);
        const unsigned half = lane < 32 ? (unsigned)bal : (unsigned)(bal >> 32);
        return (half >> half_shift) & grp_mask;
    };
    int cnt = 0;
    // the step's list takes the candidate's entry as it stands in the superset list: a fluid entry is the index difference to
    // THIS particle in both, a wall entry the wall slot
    // (2 lanes per particle: who else accepts is one DPP move away -- the ballot, the half select, the shift and the two
    //  popcounts were 9 of the ~55 vector instructions of a candidate in a pass that is ALU-bound; unsigned row arithmetic:
    //  m / LPP on a signed m costs a sign fix-up per use)
    auto push = [&](bool acc, int entry, bool wall) {
        int below, all;
        if constexpr (LPP == 2) {
            const int mine = acc ? 1 : 0, other = pair_partner(mine);
            below = sub ? other : 0;
            all = mine + other;
        } else {
            const unsigned grp = LPP == 1 ? (acc ? 1u : 0u) : group_bits(acc);
            below = __popc(grp & below_me);
            all = __popc(grp);
        }
        if (acc) {
            const unsigned m = (unsigned)(cnt + below), row = m / LPP, col = (unsigned)row_base + m % LPP;
            if ((int)row < t.nl_cap) {
                if (wall) t.nl_idx[(size_t)row * t.nl_stride + col] = entry;
                else put_delta(t.nl_pk, t.nl_stride, (int)row, (int)col, entry);
            }
        }
        cnt += all;
    };
    const int ns = active ? list_rows(spacked) : 0;
    // group-uniform trip counts: lane 0 of the group owns the most rows, its last lane the fewest fluid rows
    const int rows_all = group_extreme<LPP, true>(ns);
    const int my_fl = active ? list_fluid_rows(spacked) : 0;
    const int rows_fl = group_extreme<LPP, false>(my_fl);
    const double xi = pi.x, yi = pi.y;
    double s_in = 0.0, s_ct = 0.0;
    // Three rows per turn.  A row is entry -> position -> test: the entries (16-bit index differences, two rows per word) run
    // ahead by whole words, the positions three rows (each requested into the register whose value has just been used -- a
    // rotating buffer would make every turn wait for the newest request); requests past the last row repeat it: no branch
    // around the loads, the values are not used.
    auto fluid_rows = [&](auto fold, auto near) {
        if (rows_fl <= 0) return;
        const int last = rows_fl - 1;
        constexpr bool kFold = decltype(fold)::value;
        constexpr bool kNear = decltype(near)::value;  // every entry of the wavefront names a staged slot (kFarTileBit)
        // (clamped by this lane's last word.  Clamping by the array instead makes the word's row x stride scalar arithmetic -- four
        //  64-bit multiply-adds less per turn -- and was measured slower, 378 -> 387 us at 6 M particles: the words behind a
        //  lane's last one are then real loads of lines nobody needs; profiles/r04_scalar_words_walk_c5.txt)
        auto word = [&](int p) { return t.sl_pk[(size_t)min(p, last >> 1) * t.nl_stride + tid]; };
        auto index = [&](int d) { return kFold ? wrap_index(i + d, n_now) : i + d; };
        auto row = [&](int d, const double2 &pj) {
            double dx = xi - pj.x;
            if (kFold) dx = min_image(g, dx);
            const double dy = yi - pj.y;
            const double r2 = dx * dx + dy * dy;
            const bool acc = r2 > kR2Min && r2 < ph.kc.rcut2;
            const double W = spline_W_sel(ph.kc, r2 * rsqrt_nr(r2));  // (NaN for a coincident pair: discarded by the select)
            s_in += acc ? W : 0.0;
            push(acc, d, false);
        };
        // words: [wa wb] hold rows m .. m+3 of the current turn and the next, [wc wd] the four behind them
        int wa = w_first0, wb = w_first1, wc = word(2), wd = word(3);
        if (TILE > 0) {  // positions from the tile (a few cycles away): only the entries run ahead
            auto fetch = [&](int e) -> double2 {  // (early return, not if/else: see k_kgc_w)
                if (CODED) {
#if SPHX_EXP_PRETEND_COMPLETE_TILE & 1
                    return lds_double2(c_pos, min(e, kSlotCodes - 1));  // MEASUREMENT ONLY, see k_forces_w
#endif
                    if (kNear) return lds_double2(c_pos, e);
                    if (e < kSlotCodes) return lds_double2(c_pos, e);
                    return s.pos[wrap_index(i + e - kCodeBias, n_now)];
                }
                const int k = index(e), slot = tm.slot(k);
                if (slot >= 0) return lds_double2(c_pos, slot);
                return s.pos[k];
            };
            // (measured on the near walk at 6 M particles and not kept, 375-383 us either way: the turn's words through a walking
            //  pointer instead of word()'s two 64-bit multiply-adds; a turn's four tile reads requested together instead of one
            //  LDS round trip at the head of every row -- profiles/r04_walking_pointer_walk_c5.txt, r04_batched_lds_walk_c5.txt)
            for (int m = 0; m < rows_fl; m += 4) {
                const int we = word((m >> 1) + 4), wf = word((m >> 1) + 5);
                const int d0 = CODED ? code_lo(wa) : delta_lo(wa), d1 = CODED ? code_hi(wa) : delta_hi(wa);
                const int d2 = CODED ? code_lo(wb) : delta_lo(wb), d3 = CODED ? code_hi(wb) : delta_hi(wb);
                row(d0, fetch(d0));
                if (m + 1 < rows_fl) row(d1, fetch(d1));
                if (m + 2 < rows_fl) row(d2, fetch(d2));
                if (m + 3 < rows_fl) row(d3, fetch(d3));
                wa = wc; wb = wd; wc = we; wd = wf;
            }
            return;
        }
        // four rows per turn, their positions requested one turn ahead.  Requests past the last row decode whatever the word
        // holds -- an index difference some earlier list left there: a slot of the arrays, clamped to be sure -- so there
        // is no branch and no select around the loads; the values are not used.
        auto ahead = [&](int d) { return kFold ? wrap_index(i + d, n_now) : max(0, min(i + d, t.cap - 1)); };
        int d0 = delta_lo(wa), d1 = delta_hi(wa), d2 = delta_lo(wb), d3 = delta_hi(wb);
        double2 p0 = s.pos[index(d0)], p1 = s.pos[ahead(d1)], p2 = s.pos[ahead(d2)], p3 = s.pos[ahead(d3)];
        for (int m = 0; m < rows_fl; m += 4) {
            const int we = word((m >> 1) + 4), wf = word((m >> 1) + 5);
            const int n0 = delta_lo(wc), n1 = delta_hi(wc), n2 = delta_lo(wd), n3 = delta_hi(wd);
            row(d0, p0);
            p0 = s.pos[ahead(n0)];
            if (m + 1 < rows_fl) row(d1, p1);
            p1 = s.pos[ahead(n1)];
            if (m + 2 < rows_fl) row(d2, p2);
            p2 = s.pos[ahead(n2)];
            if (m + 3 < rows_fl) row(d3, p3);
            p3 = s.pos[ahead(n3)];
            d0 = n0; d1 = n1; d2 = n2; d3 = n3;
            wa = wc; wb = wd; wc = we; wd = wf;
        }
    };
    const bool all_near = CODED && TILE > 0 && !__any(spacked & kFarTileBit);  // (see kFarTileBit)
    if (__any(active && near_seam(g, xi))) fluid_rows(std::true_type{}, std::false_type{});
    else if (CODED && all_near) fluid_rows(std::false_type{}, std::true_type{});
    else fluid_rows(std::false_type{}, std::false_type{});
    int cnt_fl = cnt;
    for (int m = rows_fl; m < rows_all; ++m) {  // the mixed row and the wall rows
        bool acc = false, wall = false;
        int e = 0;
        if (m < ns) {
            wall = m >= my_fl;  // a lane's fluid rows come first
            double2 pj;
            if (wall) {
                e = t.sl_idx[(size_t)m * t.nl_stride + tid];
                pj = w.pos[e & (kWallBit - 1)];
            } else {
                e = CODED ? code_of_row(t.sl_pk, t.nl_stride, m, tid) : delta_of_row(t.sl_pk, t.nl_stride, m, tid);
                pj = s.pos[CODED ? coded_index(tm, e, i, n_now) : wrap_index(i + e, n_now)];
            }
            const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
            const double r2 = dx * dx + dy * dy;
            if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                acc = true;
                const double W = spline_W_sel(ph.kc, r2 * rsqrt_nr(r2));
                if (wall) s_ct += W * w.a[e & (kWallBit - 1)].x;
                else s_in += W;
            }
        }
        cnt_fl += LPP == 1 ? ((acc && !wall) ? 1 : 0) : __popc(group_bits(acc && !wall));
        push(acc, e, wall);
    }
    if (active && cnt > t.nl_cap * LPP) { atomicOr(t.flags, 1); cnt = t.nl_cap * LPP; }
    cnt_fl = min(cnt_fl, cnt);
    if (tid < t.nl_stride)
        t.nl_cnt[tid] = (cnt > sub ? (cnt - sub + LPP - 1) / LPP : 0) | ((cnt_fl > sub ? (cnt_fl - sub + LPP - 1) / LPP : 0) << 16) |
                        (CODED ? spacked & (kFarForcesBit | kFarTileBit) : 0);
    s_in = group_sum<LPP>(s_in);
    s_ct = group_sum<LPP>(s_ct);
    if (active && sub == 0) {
        const double rho = density_from_sigma(ph.w0 + s_in, s_ct, mass_i, ph.rho0, ph.inv_sigma0);
        double rhoh = rho + 0.5 * dt * drho_i;
        if (rhoh < 1e-10) rhoh = ph.rho0;
        t.a[i] = half ? make_double4(mass_i / rho, eos_pressure(rhoh, ph.rho0, ph.p0), rhoh, rho)
                      : make_double4(mass_i / rho, 0.0, 0.0, rho);
        t.vol[i] = mass_i / rho;
    }
}

// pass A, sweeping the cells (see density_body MODE 0 / 1), large-channel form.  At a few lanes per particle a lane tests
// 20-40 candidates: in the compact form every trip was a load followed by a wait for it.  Here the candidates' addresses
// -- three contiguous index ranges, known before the loop -- are requested two trips ahead, the minimum-image fold runs
// only in wavefronts near the periodic seam, the kernel value is branch-free, and kernel value and list stores are spent on
// the candidates inside the radius only (two phases, see fluid_sweep).
// (CODED: fluid entries are written slot-coded, see kSlotCodes)
template <int LPP, int MODE, bool CODED = false>
__device__ __forceinline__ void density_sweep_body_w(const Clock *clk, int q, const Grid &g, const Phys &ph, const FluidSet &s,
                                                     const FluidTmp &t, const Walls &w, int bid, int nblk, bool half)
{
    static_assert(MODE == 0 || MODE == 1, "sweeping forms only");
    __shared__ int far_parked[CODED ? kBlock : 1];
    SPHX_PASS_INDEX_AT(bid, nblk);
    if (beyond_population<LPP>(clk, t, blk)) return;
    const double2 pi = in_cap ? s.pos[i] : make_double2(0.0, 0.0);
    const int ci = in_cap ? s.cell[i] : 0;
    const bool lead = in_cap && sub == 0;
    const double mass_i = lead ? s.mass[i] : 1.0, drho_i = lead ? s.drho[i] : 0.0;
    const double dt = clk->dt;
    if (!clk->run[q]) return;
    const int n_now = clk->n;
    const bool active = i < n_now;
    constexpr bool record = MODE == 1;
    // the workgroup's tile layout, the one its passes stage until the next re-binning (tile_map_of; CODED: the lists name its slots)
    TileMap layout{0, 0, 0, 0, 0, 0};
    if (t.tmap != nullptr) {
        layout = tile_ranges<LPP>(g, s, blk, n_now, kSlotCodes);
        if (threadIdx.x == 0) store_tile_map(t, blk, layout);
    }
    double s_in = 0.0, s_ct = 0.0;
    int cnt = 0, scnt = 0;
    const int lane = threadIdx.x & 63, gbase = lane & ~(LPP - 1);
    const int row_base = tid - sub;
    const int half_shift = gbase & 31;
    const unsigned grp_mask = LPP >= 32 ? 0xffffffffu : ((1u << (LPP & 31)) - 1u);
    const unsigned below_me = (1u << sub) - 1u;
    auto group_bits = [&](bool acc) -> unsigned {
        const unsigned long long bal = __ballot(acc);
        const unsigned half_ = lane < 32 ? (unsigned)bal : (unsigned)(bal >> 32);
        return (half_ >> half_shift) & grp_mask;
    };
    // fluid entries go to the packed list as index differences to this group's particle, wall entries to the 32-bit one
    auto push_to = [&](int *idx, int *pk, int cap, int &n, bool acc, int entry, bool wall) {  // (see push of density_walk_body)
        int below, all;
        if constexpr (LPP == 2) {
            const int mine = acc ? 1 : 0, other = pair_partner(mine);
            below = sub ? other : 0;
            all = mine + other;
        } else {
            const unsigned grp = LPP == 1 ? (acc ? 1u : 0u) : group_bits(acc);
            below = __popc(grp & below_me);
            all = __popc(grp);
        }
        if (acc) {
            const unsigned m = (unsigned)(n + below), row = m / LPP, col = (unsigned)row_base + m % LPP;
            if ((int)row < cap) {
                if (wall) idx[(size_t)row * t.nl_stride + col] = entry;
                else put_delta(pk, t.nl_stride, (int)row, (int)col, entry);
            }
        }
        n += all;
    };
    const double xi = pi.x, yi = pi.y;
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, wlo[3] = {0, 0, 0}, whi[3] = {0, 0, 0};
    bool near_wall = false;
    if (active) {
        const int cx = ci / g.ncy, cy = ci - cx * g.ncy;
        const int cylo = max(cy - 1, 0), cyhi = min(cy + 1, g.ncy - 1);
        near_wall = w.row_any[cy] != 0;
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox) {
            int col = cx + ox;
            bool ok = true;
            if (g.periodic) {
                if (duplicate_column(g, ox)) ok = false;
                if (col < 0) col += g.ncx;
                else if (col >= g.ncx) col -= g.ncx;
            } else if (col < 0 || col >= g.ncx) {
                ok = false;
            }
            const int c0 = col * g.ncy + cylo, c1 = col * g.ncy + cyhi + 1;
            lo[ox + 1] = ok ? s.start[c0] : 0;
            hi[ox + 1] = ok ? s.start[c1] : 0;
            wlo[ox + 1] = ok ? w.start[c0] : 0;
            whi[ox + 1] = ok ? w.start[c1] : 0;
        }
    }
    // the three ranges as ONE virtual index range (see density_body); a group's lanes take LPP consecutive candidates per trip
    const int n0 = hi[0] - lo[0], n1 = hi[1] - lo[1], n2 = hi[2] - lo[2];
    const int nfl = n0 + n1 + n2;
    // v -> particle index: two independent selects (a nested choice of three becomes branches, or a table in scratch memory)
    const int n01 = n0 + n1, off2 = lo[2] - n01, d12 = (lo[1] - n0) - off2, d01 = lo[0] - (lo[1] - n0);
    auto fluid_index = [&](int v) { return v + off2 + (v < n01 ? d12 : 0) + (v < n0 ? d01 : 0); };
    // kFarTileBit / kFarForcesBit from the candidate ranges rather than from the entries handed out (a running maximum is one
    // register more than this kernel has for six waves -- and so is this word, which waits in LDS until the counts are stored):
    // each of the particle's three ranges must lie inside the layout's range of that column.  The lanes of a group agree; which of them stores an entry and which one walks it differ, but they sit in
    // one wavefront, and wavefronts decide.
    if (CODED) {
        // (the particle's left / own / right range against the layout's second / first / third: a particle of the column the
        //  layout was made for; any other one -- the tail of a workgroup that runs into the next column -- fails the test)
        const bool miss = (n0 > 0 && (lo[0] < layout.lo1 || hi[0] > layout.lo1 + layout.len1)) ||
                          (n1 > 0 && (lo[1] < layout.lo0 || hi[1] > layout.lo0 + layout.len0)) ||
                          (n2 > 0 && (lo[2] < layout.lo2 || hi[2] > layout.lo2 + layout.len2));
        const int top = n2 > 0 ? layout.len0 + layout.len1 + (hi[2] - layout.lo2) : (n0 > 0 ? layout.len0 + (hi[0] - layout.lo1) : hi[1] - layout.lo0);
        far_parked[threadIdx.x] = miss ? (kFarForcesBit | kFarTileBit) : (top > kForceSlots ? kFarForcesBit : 0);
    }
    // Two phases per chunk of 64 trips.  Only a third of the candidates of the 3 x 3 cells lie inside the (superset) radius,
    // and kernel value + packed list stores are four fifths of a trip's instructions: phase 1 only measures distances and
    // marks the trips inside the radius in a bit mask (positions requested three trips ahead, each into a register of its
    // own -- a rotating buffer would make every turn wait for the newest request; requests past the end repeat the last
    // candidate: no branch around the load); phase 2 visits the marked trips only.  A group's lanes mark different
    // trips, so its list holds the same entries as the compact form's in a slightly different (still fixed) order.
    const double r2_mark = record ? t.sl_rcut2 : ph.kc.rcut2;
    auto fluid_sweep = [&](auto fold) {
        if (nfl <= 0) return;
        auto index_at = [&](int v) { return fluid_index(min(v, nfl - 1)); };
        auto dist2 = [&](const double2 &pj) {
            double dx = xi - pj.x;
            if (decltype(fold)::value) dx = min_image(g, dx);
            const double dy = yi - pj.y;
            return dx * dx + dy * dy;
        };
        for (int base = 0; base < nfl; base += 64 * LPP) {
            const int n_tr = min((nfl - base + LPP - 1) / LPP, 64);  // trips of this chunk (the same for the group's lanes)
            unsigned long long mask = 0;
            auto mark = [&](int tt, const double2 &pj) {
                const double r2 = dist2(pj);
                const bool inside = base + tt * LPP + sub < nfl && r2 > kR2Min && r2 < r2_mark;
                mask |= (unsigned long long)(inside ? 1 : 0) << tt;
            };
            double2 pa = s.pos[index_at(base + sub)], pb = s.pos[index_at(base + LPP + sub)],
                    pc = s.pos[index_at(base + 2 * LPP + sub)];
            for (int tt = 0; tt < n_tr; tt += 3) {
                mark(tt, pa);
                pa = s.pos[index_at(base + (tt + 3) * LPP + sub)];
                if (tt + 1 < n_tr) mark(tt + 1, pb);
                pb = s.pos[index_at(base + (tt + 4) * LPP + sub)];
                if (tt + 2 < n_tr) mark(tt + 2, pc);
                pc = s.pos[index_at(base + (tt + 5) * LPP + sub)];
            }
            int rounds = __popcll(mask);
#pragma unroll
            for (int off = LPP / 2; off > 0; off >>= 1) rounds = max(rounds, __shfl_xor(rounds, off));
            // (the position of the next marked trip is requested before the current one is worked on)
            auto next_marked = [&](bool &has, int &k) {
                has = mask != 0;
                const int tt = has ? __ffsll((long long)mask) - 1 : 0;
                mask &= mask - 1;
                k = index_at(base + tt * LPP + sub);
            };
            bool has, has_n;
            int k, k_n;
            next_marked(has, k);
            double2 pj = s.pos[k];
            for (int r = 0; r < rounds; ++r) {
                next_marked(has_n, k_n);
                const double2 pj_n = s.pos[k_n];
                const double r2 = dist2(pj);
                const bool acc = has && r2 < ph.kc.rcut2;
                const double W = spline_W_sel(ph.kc, r2 * rsqrt_nr(r2));
                s_in += acc ? W : 0.0;
                int d = 0;
                if (has) {
                    const int slot = CODED ? layout.slot(k) : -1;
                    d = slot >= 0 ? slot
                                  : encode_delta(k, i, n_now, g.periodic != 0, t.flags, CODED ? kCodedDeltaMax : kDeltaMax) + (CODED ? kCodeBias : 0);
                }
                push_to(t.nl_idx, t.nl_pk, t.nl_cap, cnt, acc, d, false);
                if (record) push_to(t.sl_idx, t.sl_pk, t.sl_cap, scnt, has, d, false);
                has = has_n; k = k_n; pj = pj_n;
            }
        }
    };
    if (__any(active && near_seam(g, xi))) fluid_sweep(std::true_type{});
    else fluid_sweep(std::false_type{});
    int cnt_fl = cnt, scnt_fl = scnt;
    if (near_wall) {
        const int w0 = whi[0] - wlo[0], w1 = whi[1] - wlo[1], w2 = whi[2] - wlo[2];
        const int nwl = w0 + w1 + w2;
        auto wall_index = [&](int v) { return v < w0 ? wlo[0] + v : (v < w0 + w1 ? wlo[1] + (v - w0) : wlo[2] + (v - w0 - w1)); };
        if (nwl > 0) {
            int ka = wall_index(min(sub, nwl - 1));
            double2 pa = w.pos[ka];
            double va = w.a[ka].x;
            for (int vb = 0; vb < nwl; vb += LPP) {
                const int kb = wall_index(min(vb + LPP + sub, nwl - 1));
                const double2 pb = w.pos[kb];
                const double vb_ = w.a[kb].x;
                const double dx = min_image(g, xi - pa.x), dy = yi - pa.y;
                const double r2 = dx * dx + dy * dy;
                const bool in = vb + sub < nwl && r2 > kR2Min;
                const bool acc = in && r2 < ph.kc.rcut2;
                const double W = spline_W_sel(ph.kc, r2 * rsqrt_nr(r2));
                s_ct += acc ? W * va : 0.0;
                push_to(t.nl_idx, t.nl_pk, t.nl_cap, cnt, acc, ka | kWallBit, true);
                if (record) push_to(t.sl_idx, t.sl_pk, t.sl_cap, scnt, in && r2 < t.sl_rcut2, ka | kWallBit, true);
                ka = kb; pa = pb; va = vb_;
            }
        }
    }
    if (active && cnt > t.nl_cap * LPP) { atomicOr(t.flags, 1); cnt = t.nl_cap * LPP; }
    cnt_fl = min(cnt_fl, cnt);
    if (record && active && scnt > t.sl_cap * LPP) { atomicOr(t.flags, 1); scnt = t.sl_cap * LPP; }
    scnt_fl = min(scnt_fl, scnt);
    const int far_bits = CODED ? far_parked[threadIdx.x] : 0;
    if (tid < t.nl_stride)
        t.nl_cnt[tid] = (cnt > sub ? (cnt - sub + LPP - 1) / LPP : 0) | ((cnt_fl > sub ? (cnt_fl - sub + LPP - 1) / LPP : 0) << 16) | far_bits;
    if (record && tid < t.nl_stride)
        t.sl_cnt[tid] = (scnt > sub ? (scnt - sub + LPP - 1) / LPP : 0) | ((scnt_fl > sub ? (scnt_fl - sub + LPP - 1) / LPP : 0) << 16) | far_bits;
    s_in = group_sum<LPP>(s_in);
    s_ct = group_sum<LPP>(s_ct);
    if (active && sub == 0) {
        const double rho = density_from_sigma(ph.w0 + s_in, s_ct, mass_i, ph.rho0, ph.inv_sigma0);
        double rhoh = rho + 0.5 * dt * drho_i;
        if (rhoh < 1e-10) rhoh = ph.rho0;
        t.a[i] = half ? make_double4(mass_i / rho, eos_pressure(rhoh, ph.rho0, ph.p0), rhoh, rho)
                      : make_double4(mass_i / rho, 0.0, 0.0, rho);
        t.vol[i] = mass_i / rho;
    }
}

// n_tiles: workgroup-sized tiles of the pass; the grid may be smaller (grid-stride over the tiles: the conditional launches
// of a dynamic context, which are idle most of the time, see launch_physics)
template <int LPP, int MODE, bool CODED = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(CODED ? 6 : 5))) void k_density_sweep_w(const Clock *clk, int q, Grid g, Phys ph, FluidSet s,
                                                            FluidTmp t, Walls w, int cond_fresh, int n_tiles)
{
    if (cond_fresh >= 0 && (clk->fresh != 0) != (cond_fresh != 0)) return;
    for (int b = (int)blockIdx.x; b < n_tiles; b += (int)gridDim.x)
        density_sweep_body_w<LPP, MODE, CODED>(clk, q, g, ph, s, t, w, b, n_tiles, true);
}

template <int LPP, int TILE = 0, bool CODED = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((TILE > 0 || LPP <= 2) ? 8 : 1))) void k_density_w(const Clock *clk, int q, Grid g, Phys ph, FluidSet s,
                                                      FluidTmp t, Walls w, int cond_fresh)
{
    constexpr int kSlots = TILE > 0 ? TILE : 1;
    __shared__ double2 c_pos[kSlots];
    const int part = cond_fresh >= 0 ? cond_fresh >> kPassPartShift : 0;  // (see slab_part_skips)
    if (part) cond_fresh &= (1 << kPassPartShift) - 1;
    if (cond_fresh >= 0 && (clk->fresh != 0) != (cond_fresh != 0)) return;
    if (part && slab_part_skips<LPP>(g, s, xcd_block((int)blockIdx.x, (int)gridDim.x), part)) return;
    density_walk_body<LPP, TILE, CODED>(clk, q, g, ph, s, t, w, (int)blockIdx.x, (int)gridDim.x, true, c_pos);
}

// pass B (see k_kgc); CODED: slot-coded list entries (kSlotCodes)
template <int LPP, int TILE, bool CODED = false>
__global__ __launch_bounds__(kBlock) void k_kgc_w(const Clock *clk, int q, Grid g, Phys ph, FluidSet s,
                                                  FluidTmp t, Walls w, int finish_half)
{
    static_assert(!CODED || TILE == kSlotCodes, "slot-coded lists: this pass stages the whole layout");
    constexpr int kSlots = TILE > 0 ? TILE : 1;
    __shared__ double2 c_pos[kSlots];
    __shared__ double c_vol[kSlots];
    SPHX_PASS_INDEX();
    if (beyond_population<LPP>(clk, t, blk)) return;
    const double2 pi = in_cap ? s.pos[i] : make_double2(0.0, 0.0);
    const bool closes = finish_half && in_cap && sub == 0;  // (see k_kgc)
    const double4 a_own = closes ? t.a[i] : make_double4(0.0, 0.0, 0.0, 0.0);
    const double drho_own = closes ? s.drho[i] : 0.0;
    const int packed = t.nl_cnt[tid];
    const int w0 = t.nl_pk[tid], w1 = t.nl_pk[(size_t)t.nl_stride + tid];
    const TileMap layout = TILE > 0 ? tile_map_of(t, blk) : TileMap{0, 0, 0, 0, 0, 0};
    if (!clk->run[q]) return;
    const int n_now = clk->n;
    const bool active = i < n_now;
    TileMap tm{0, 0, 0, 0, 0, 0};
    if (TILE > 0) {
        tm = staged_map<LPP>(layout, blk, n_now, TILE);
        static_assert(TILE <= 2 * kBlock, "two slots per thread");
        {  // (both of a thread's slots requested before either is waited for: see k_forces_w)
            const int total = tm.total(), sl0 = threadIdx.x, sl1 = threadIdx.x + kBlock;
            const bool h0 = sl0 < total, h1 = sl1 < total;
            const int k0 = tm.index(h0 ? sl0 : 0) + n2_prologue_cost(sl0, tm), k1 = tm.index(h1 ? sl1 : 0) + n2_prologue_cost(sl1, tm);
            double2 p0, p1;
            double u0, u1;
            if (h0) { p0 = s.pos[k0]; u0 = t.vol[k0]; }
            if (h1) { p1 = s.pos[k1]; u1 = t.vol[k1]; }
            if (h0) { c_pos[sl0] = p0; c_vol[sl0] = u0; }
            if (h1) { c_pos[sl1] = p1; c_vol[sl1] = u1; }
        }
        __syncthreads();
    }
    // (early return, not if/else: the compiler would merge the two arms into one FLAT load through a selected pointer)
    auto fetch = [&](int k, double2 &pj, double &Volj) {
        if (TILE > 0) {
            const int slot = tm.slot(k);
            if (slot >= 0) { pj = c_pos[slot]; Volj = c_vol[slot]; return; }
        }
        pj = s.pos[k]; Volj = t.vol[k];
    };
    const int rows = active ? list_rows(packed) : 0, rows_fl = active ? list_fluid_rows(packed) : 0;
    const double xi = pi.x, yi = pi.y;
    double a11 = 0.0, a12 = 0.0, a21 = 0.0, a22 = 0.0;
    auto term = [&](double dx, double dy, double Volj) {
        const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
        const double ex = dx * inv_r, ey = dy * inv_r;
        const double fxj = spline_dW_in(ph.kc, r) * Volj;
        a11 -= dx * (fxj * ex);
        a12 -= dx * (fxj * ey);
        a21 -= dy * (fxj * ex);
        a22 -= dy * (fxj * ey);
    };
    auto fetch_code = [&](int e, double2 &pj, double &Volj) {  // (CODED; early return: see above)
        // (LDS-qualified reads: with plain ones the compiler folds both arms into FLAT loads through selected pointers)
#if SPHX_EXP_PRETEND_COMPLETE_TILE & 2
        e = min(e, kSlotCodes - 1);  // MEASUREMENT ONLY, see k_forces_w
#endif
        if (e < kSlotCodes) { pj = lds_double2(c_pos, e); Volj = lds_double(c_vol, e); return; }
        const int k = wrap_index(i + e - kCodeBias, n_now);
        pj = s.pos[k]; Volj = t.vol[k];
    };
    const bool seam = __any(active && near_seam(g, xi));
    const bool all_near = CODED && !__any(packed & kFarTileBit);  // (see kFarTileBit)
    if (CODED && seam)
        walk_fluid_rows<true>(t, tid, i, rows_fl, w0, w1, [&](int e) {
            double2 pj; double Volj;
            fetch_code(e, pj, Volj);
            term(min_image(g, xi - pj.x), yi - pj.y, Volj);
        });
    else if (CODED && all_near)
        walk_fluid_rows<true>(t, tid, i, rows_fl, w0, w1, [&](int e) {
            const double2 pj = lds_double2(c_pos, e);
            term(xi - pj.x, yi - pj.y, lds_double(c_vol, e));
        });
    else if (CODED)
        walk_fluid_rows<true>(t, tid, i, rows_fl, w0, w1, [&](int e) {
            double2 pj; double Volj;
            fetch_code(e, pj, Volj);
            term(xi - pj.x, yi - pj.y, Volj);
        });
    else if (seam)
        walk_fluid_rows(t, tid, i, rows_fl, w0, w1, [&](int k) {
            double2 pj; double Volj;
            fetch(wrap_index(k, n_now), pj, Volj);
            term(min_image(g, xi - pj.x), yi - pj.y, Volj);
        });
    else
        walk_fluid_rows(t, tid, i, rows_fl, w0, w1, [&](int k) {
            double2 pj; double Volj;
            fetch(k, pj, Volj);
            term(xi - pj.x, yi - pj.y, Volj);
        });
    walk_wall_rows(t, tid, rows_fl, rows, [&](int k) {
        const double2 pj = w.pos[k];
        term(min_image(g, xi - pj.x), yi - pj.y, w.a[k].x);
    });
    a11 = group_sum<LPP>(a11);
    a12 = group_sum<LPP>(a12);
    a21 = group_sum<LPP>(a21);
    a22 = group_sum<LPP>(a22);
    if (active && sub == 0) {
        const Mat2 B = kgc_from_A(a11, a12, a21, a22);
        t.B[i] = make_double4(B.m11, B.m12, B.m21, B.m22);
        if (finish_half) {
            double rhoh, p_half;
            half_state(ph, a_own.w, drho_own, clk->dt, rhoh, p_half);
            t.a[i] = make_double4(a_own.x, p_half, rhoh, a_own.w);
        }
    }
}

// pass CD (see k_forces); TILE > 0: neighbour records come from the LDS tile (tile_ranges)
struct FluidNb {
    double2 p, v;
    double4 a, B;
};

// (CODED: slot-coded list entries, see kSlotCodes -- this pass's tile holds the first TILE slots of the layout)
template <int LPP, int TILE, bool CODED = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TILE > 400 ? 4 : 5))) void k_forces_w(const Clock *clk, int q, Grid g, Phys ph, FluidSet s,
                                                     FluidTmp t, Walls w)
{
    static_assert(!CODED || (TILE > 0 && TILE <= kSlotCodes), "slot-coded lists need a tile");
    constexpr int kSlots = TILE > 0 ? TILE : 1;
    // (88 bytes per staged neighbour: of {Vol, p_half, rho_half, rho} the pass reads three)
    __shared__ double2 c_pos[kSlots], c_vel[kSlots], c_vp[kSlots];
    __shared__ double c_rh[kSlots];
    __shared__ double4 c_B[kSlots];
    SPHX_PASS_INDEX();
    if (beyond_population<LPP>(clk, t, blk)) {  // (a slab's arrays have 15-20 % of slack: see beyond_population)
        if (threadIdx.x == 0 && clk->run[q]) t.dpart[blk] = 0.0;
        return;
    }
    const double2 pi = in_cap ? s.pos[i] : make_double2(0.0, 0.0);
    const double2 vi = in_cap ? s.vel[i] : make_double2(0.0, 0.0);
    const double4 ai = in_cap ? t.a[i] : make_double4(1.0, 0.0, 0.0, 0.0);
    const double4 Bi = in_cap ? t.B[i] : make_double4(1.0, 0.0, 0.0, 1.0);
    const double mi = in_cap ? s.mass[i] : 1.0;
    const int packed = t.nl_cnt[tid];
    const int w0 = t.nl_pk[tid], w1 = t.nl_pk[(size_t)t.nl_stride + tid];
    const bool tracked = s.posb != nullptr;
    const double2 pb = (tracked && in_cap && sub == 0) ? s.posb[i] : make_double2(0.0, 0.0);
    const TileMap layout = TILE > 0 ? tile_map_of(t, blk) : TileMap{0, 0, 0, 0, 0, 0};
    const double dt = clk->dt;
    const bool want_out = step_outputs_wanted(clk, t);
    if (!clk->run[q]) return;
    const int n_now = clk->n;
    const bool active = i < n_now;
    const int rows = active ? list_rows(packed) : 0, rows_fl = active ? list_fluid_rows(packed) : 0;
    const double h = ph.kc.h, soft = 0.01 * h;
    double ax = 0.0, ay = 0.0, ix = 0.0, iy = 0.0, px = 0.0, py = 0.0, d2 = 0.0;
    const double xi = pi.x, yi = pi.y, vxi = vi.x, vyi = vi.y;
    const double Voli = ai.x, p_i = ai.y, rhoh_i = ai.z;
    const double b11i = Bi.x, b12i = Bi.y, b21i = Bi.z, b22i = Bi.w;

    TileMap tm{0, 0, 0, 0, 0, 0};
    if (TILE > 0) {
        tm = staged_map<LPP>(layout, blk, n_now, TILE);
        // (a thread stages up to two slots: both requested before either is waited for -- as a loop, the second trip's
        //  loads left only after the first trip's had come back and gone to LDS: a memory round trip more in the prologue
        //  every workgroup of the CU waits through)
        static_assert(TILE <= 2 * kBlock, "two slots per thread");
        {
            const int total = tm.total(), sl0 = threadIdx.x, sl1 = threadIdx.x + kBlock;
            const bool h0 = sl0 < total, h1 = sl1 < total;
            const int k0 = tm.index(h0 ? sl0 : 0) + n2_prologue_cost(sl0, tm), k1 = tm.index(h1 ? sl1 : 0) + n2_prologue_cost(sl1, tm);
            double2 p0, v0, p1, v1;
            double4 a0, B0, a1, B1;
            if (h0) { p0 = s.pos[k0]; v0 = s.vel[k0]; a0 = t.a[k0]; B0 = t.B[k0]; }
            if (h1) { p1 = s.pos[k1]; v1 = s.vel[k1]; a1 = t.a[k1]; B1 = t.B[k1]; }
            if (h0) { c_pos[sl0] = p0; c_vel[sl0] = v0; c_vp[sl0] = make_double2(a0.x, a0.y); c_rh[sl0] = a0.z; c_B[sl0] = B0; }
            if (h1) { c_pos[sl1] = p1; c_vel[sl1] = v1; c_vp[sl1] = make_double2(a1.x, a1.y); c_rh[sl1] = a1.z; c_B[sl1] = B1; }
        }
        __syncthreads();
    }
    const int n_staged = tm.total();
    auto fetch_code = [&](int e) {  // (CODED)
        FluidNb n;
#if SPHX_EXP_PRETEND_COMPLETE_TILE & 4
        // MEASUREMENT ONLY (tools/probes/build_variant_lib.sh -DSPHX_EXP_PRETEND_COMPLETE_TILE=<1 pass A | 2 B | 4 CD | 8 E>, never
        // in libsphx.so): what would a particle order buy whose
        // neighbourhoods fit TILE slots?  Entries the tile does not hold read some staged slot instead of global memory: the
        // results are wrong, the instruction and memory streams are those of a complete tile (timed with sphx_ctx_time_kernel,
        // which does not feed the outputs back into the state).
        e = e < n_staged ? e : (e & 255);
#endif
        if (e < n_staged) {  // (LDS-qualified reads: see k_kgc_w)
            const double2 vp = lds_double2(c_vp, e);
            n.p = lds_double2(c_pos, e); n.v = lds_double2(c_vel, e); n.a = make_double4(vp.x, vp.y, lds_double(c_rh, e), 0.0); n.B = lds_double4(c_B, e);
            return n;
        }
        const int k = coded_index(layout, e, i, n_now);
        n.p = s.pos[k]; n.v = s.vel[k]; n.a = t.a[k]; n.B = t.B[k];
        return n;
    };
    auto fetch = [&](int k) {
        FluidNb n;
        if (TILE > 0) {  // (early return, not if/else: see k_kgc_w)
            const int slot = tm.slot(k);
            if (slot >= 0) {
                const double2 vp = c_vp[slot];
                n.p = c_pos[slot]; n.v = c_vel[slot]; n.a = make_double4(vp.x, vp.y, c_rh[slot], 0.0); n.B = c_B[slot];
                return n;
            }
        }
        n.p = s.pos[k]; n.v = s.vel[k]; n.a = t.a[k]; n.B = t.B[k];
        return n;
    };
    // The pair term, arranged for the instruction count (the pass is VALU-bound: ~70 % of its cycles issue vector
    // instructions).  Against the reference's order of operations (sph_physics_mex.c:1100-1140), equal up to rounding:
    //   * un_l - un_r = (v_i - v_j) . e, with the velocity difference the viscous term needs anyway;
    //   * p_face = (p_avg + p_star) / 2 with p_star = p_avg + beta rho_bar (un_l - un_r) / 2 collapses to
    //     (p_i + p_j) / 2 + [beta / 8] (rho_i + rho_j) (un_l - un_r); beta / 8 = min(3/8 max(du, 0), c_f / 8) exactly
    //     (powers of two);
    //   * mu is multiplied into the viscous sums once, after the walk.
    const double c_f8 = 0.125 * ph.c_f;
    auto fluid_pair = [&](const FluidNb &n, double dx) {
        const double dy = yi - n.p.y;
        const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
        const double ex = dx * inv_r, ey = dy * inv_r;
        const double dWVj = spline_dW_in(ph.kc, r) * n.a.x;
        const double tx = (b11i + n.B.x) * ex + (b12i + n.B.y) * ey;
        const double ty = (b21i + n.B.z) * ex + (b22i + n.B.w) * ey;
        const double eBe = ex * tx + ey * ty;
        const double dvx = vxi - n.v.x, dvy = vyi - n.v.y;
        // viscous (without mu)
        const double coeff = eBe * dWVj * rcp_nr(r + soft);
        ax += coeff * dvx;
        ay += coeff * dvy;
        // transport
        ix -= dWVj * tx;
        iy -= dWVj * ty;
        // pressure (Riemann-dissipated face pressure)
        const double du = dvx * ex + dvy * ey;
        const double beta8 = fmin(0.375 * fmax(du, 0.0), c_f8);
        const double p_face = fma(beta8 * (rhoh_i + n.a.z), du, 0.5 * (p_i + n.a.y));
        const double pw = p_face * dWVj;
        px -= pw * tx;
        py -= pw * ty;
    };
    const bool seam = __any(active && near_seam(g, xi));
    const bool all_near = CODED && TILE == kForceSlots && !__any(packed & kFarForcesBit);  // (see kFarTileBit)
    if (CODED && seam)
        walk_fluid_rows<true>(t, tid, i, rows_fl, w0, w1, [&](int e) {
            const FluidNb n = fetch_code(e);
            fluid_pair(n, min_image(g, xi - n.p.x));
        });
    else if (CODED && all_near)
        walk_fluid_rows<true>(t, tid, i, rows_fl, w0, w1, [&](int e) {
            FluidNb n;
            const double2 vp = lds_double2(c_vp, e);
            n.p = lds_double2(c_pos, e); n.v = lds_double2(c_vel, e); n.a = make_double4(vp.x, vp.y, lds_double(c_rh, e), 0.0); n.B = lds_double4(c_B, e);
            fluid_pair(n, xi - n.p.x);
        });
    else if (CODED)
        walk_fluid_rows<true>(t, tid, i, rows_fl, w0, w1, [&](int e) {
            const FluidNb n = fetch_code(e);
            fluid_pair(n, xi - n.p.x);
        });
    else if (seam)
        walk_fluid_rows(t, tid, i, rows_fl, w0, w1, [&](int k) {
            const FluidNb n = fetch(wrap_index(k, n_now));
            fluid_pair(n, min_image(g, xi - n.p.x));
        });
    else
        walk_fluid_rows(t, tid, i, rows_fl, w0, w1, [&](int k) {
            const FluidNb n = fetch(k);
            fluid_pair(n, xi - n.p.x);
        });
    // wall neighbours: viscous and transport now, pressure once force_prior is complete (p_wall needs it, :931-934)
    walk_wall_rows(t, tid, rows_fl, rows, [&](int k) {
        const double2 pj = w.pos[k];
        const double4 wj = w.a[k];
        const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
        const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
        const double ex = dx * inv_r, ey = dy * inv_r;
        const double dWVj = spline_dW_in(ph.kc, r) * wj.x;
        const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
        const double eBe = ex * tx + ey * ty;
        const double coeff = 4.0 * eBe * dWVj * rcp_nr(r + soft);  // (mu: after the walk, see fluid_pair)
        ax += coeff * (vxi - wj.y);
        ay += coeff * (vyi - wj.z);
        ix -= 2.0 * dWVj * tx;
        iy -= 2.0 * dWVj * ty;
    });
    ax = group_sum<LPP>(ax) * ph.mu;
    ay = group_sum<LPP>(ay) * ph.mu;
    ix = group_sum<LPP>(ix);
    iy = group_sum<LPP>(iy);
    const double fpx = ax * Voli + mi * ph.g;  // + gravity, SPH_Poiseuille.m:392
    const double fpy = ay * Voli;
    const double inv_m = rcp_nr(mi);
    if (rows > rows_fl) {
        const double acx = fpx * inv_m, acy = fpy * inv_m;
        walk_wall_rows(t, tid, rows_fl, rows, [&](int k) {
            const double2 pj = w.pos[k];
            const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
            const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
            const double ex = dx * inv_r, ey = dy * inv_r;
            const double dWVj = spline_dW_in(ph.kc, r) * w.a[k].x;
            const double face = -(acx * ex + acy * ey);
            const double p_wall = p_i + rhoh_i * r * fmax(0.0, face);
            const double tx = b11i * ex + b12i * ey, ty = b21i * ex + b22i * ey;
            px -= (p_i + p_wall) * dWVj * tx;
            py -= (p_i + p_wall) * dWVj * ty;
        });
    }
    px = group_sum<LPP>(px);
    py = group_sum<LPP>(py);
    if (active && sub == 0) {
        const double fx = px * Voli, fy = py * Voli;
        const double vxn = vxi + (fpx + fx) * inv_m * dt;
        const double vyn = vyi + (fpy + fy) * inv_m * dt;
        // transport limiter min(1, 100 |inc|^2 / h^2) (sph_physics_mex.c:702-710) with 1/h^2 multiplied in
        const double lim = fmin(100.0 * (ix * ix + iy * iy) * (ph.kc.inv_h * ph.kc.inv_h), 1.0);
        const double sx = ph.tc * h * h * lim * ix, sy = ph.tc * h * h * lim * iy;
        double xo = xi + sx, yo = yi + sy;
        xo += 0.5 * dt * vxi;
        yo += 0.5 * dt * vyi;
        xo += 0.5 * dt * vxn;
        yo += 0.5 * dt * vyn;
        if (tracked && (!g.own_by_cell || owns(g, 0.0, s.cell[i]))) {  // (a slab bounds the drift of what it owns)
            const double ddx = min_image(g, xo - pb.x), ddy = yo - pb.y;
            d2 = ddx * ddx + ddy * ddy;
            if (d2 != d2) d2 = INFINITY;
        }
        // periodic wrap (SPH_Poiseuille.m:570-577): a step moves a particle by a tiny fraction of DL, so x - floor(x/DL) DL
        // is x - DL, x + DL or x -- the same values without the division (a slab wraps when particles change owner)
        if (g.periodic) xo = xo >= ph.DL ? xo - ph.DL : (xo < 0.0 ? xo + ph.DL : xo);
        t.posn[i] = make_double2(xo, yo);
        t.veln[i] = make_double2(vxn, vyn);
        if (want_out) {
            t.fp[i] = make_double2(fpx, fpy);
            t.f[i] = make_double2(fx, fy);
        }
    }
    // largest drift from the binning positions (bounds how stale the cell grid may get, see Clock::drift)
    d2 = wave_max(d2);
    __shared__ double s_d2[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s_d2[threadIdx.x >> 6] = d2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_d2[0];
        for (int k = 1; k < kBlock / 64; ++k) m = fmax(m, s_d2[k]);
        t.dpart[blk] = m;
    }
}

// block-wide exclusive scan of one int per thread (NT threads); returns the block total
template <int NT>
__device__ __forceinline__ int block_exclusive_scan_t(int v, int &total, int *s_wave /*[NT/64+1]*/)
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
        const int ws = lane < NT / 64 ? s_wave[lane] : 0;
        int winc = ws;
#pragma unroll
        for (int off = 1; off < NT / 64; off <<= 1) {
            const int o = __shfl_up(winc, off);
            if (lane >= off) winc += o;
        }
        if (lane < NT / 64) s_wave[lane] = winc - ws;  // exclusive wave offsets
        if (lane == NT / 64 - 1) s_wave[NT / 64] = winc;
    }
    __syncthreads();
    const int res = s_wave[wave] + inc - v;
    total = s_wave[NT / 64];
    __syncthreads();
    return res;
}

// advance the device clock by the step that has just been computed (one thread)
// drift: largest distance from the binning positions (< 0: not tracked); rebuilt: this step ends with a fresh grid.
// dyn_K > 0: dynamic context -- decide here whether this step ends with a re-binning (K-th step, or drift bound hit)
__device__ __forceinline__ void clock_step(Clock *clk, Clock c, int q, const Phys &ph, double vmax, int flags,
                                           int n_new, double drift = -1.0, int rebuilt = 1, double half_skin = 0.0,
                                           int dyn_K = 0)
{
    if (dyn_K > 0) {
        const bool by_drift = !(drift <= half_skin);
        const bool sched = c.pos_count >= dyn_K - 1;  // pass E has taken the histogram already
        const bool rb = by_drift || sched;
        c.rebuild_now = rb ? (sched ? 2 : 1) : 0;     // 2: histogram done, k_bin skips
        c.fresh = c.rebuild_now;  // the next pass A starts from a fresh grid
        c.pos_count = rb ? 0 : c.pos_count + 1;
        if (q == 0) c.pos_q[1] = (short)c.pos_count;  // (for the NEXT slot; this slot's own copy stays as its workgroups read it)
        else c.pos_q[0] = (short)c.pos_count;
        c.drift = rb ? 0.0 : drift;
        if (by_drift) c.n_drift_rebuilds += 1;
        if (rb) c.n_rebins += 1;
    } else if (drift >= 0.0) {
        c.drift = rebuilt ? 0.0 : drift;
        // every step sweeps the cells the particles were BINNED into; that finds all neighbours only while no
        // particle has drifted more than half the skin -> stop the loop, the host re-bins and resumes
        if (!(c.drift <= half_skin)) c.need_rebuild = 1;
    }
    c.vmax = vmax;
    c.t += c.n_in > 1 ? c.dt * c.n_in : c.dt;  // SPH_Poiseuille.m:267 (dual-rate: n_in sub-steps of dt were taken)
    c.dt_last = c.dt;
    c.step += 1;
    if (c.steps_left > 0) c.steps_left -= 1;
    if (!(c.vmax == c.vmax) || isinf(c.vmax)) c.status = SPHX_ERR_DIVERGED;
    if (flags) c.status = SPHX_ERR_GRID;  // neighbour-list / slab-buffer overflow
    if (n_new >= 0) c.n = n_new;
    c.dt = next_dt(c, ph);
    const int go = loop_continues(c) ? 1 : 0;
    if (q == 0) c.run[1] = go;  // constant indices: c.run[1 - q] would force the whole struct into scratch memory --
    else c.run[0] = go;         // 120 bytes per lane of EVERY wave of pass E when the tail workgroup lives there
    *clk = c;
    if (!go && c.pub) *c.pub = c;  // the batch ends here: tell the host (see Clock::pub)
}

// ---------------------------------------------------------------------------------------------
// pass E: continuity rate with the kicked velocities (integration_2nd, sph_physics_mex.c:1076-1116),
// final half-step of rho and EOS (:1440-1450), per-block max |v|^2 over owned particles for the next
// dt (SPH_Poiseuille.m:521) and -- single-GPU -- the cell histogram of the end-of-step positions
// (neighbour rebuild, the K0 insert of mex/sph_neighbor_search_mex.c:269-296).
// ---------------------------------------------------------------------------------------------
// Small channels: the clock update rides in pass E instead of being a launch of its own.  One extra workgroup
// (the last one) waits for the per-block max |v|^2 of all the others, then advances the clock exactly as
// k_clock_scan would.  Every vpart entry is its own "ready" flag: the producers store it with an agent-scope
// atomic (past the non-coherent L2), the tail polls with agent-scope atomic loads and puts the "empty" pattern
// back for the next step.  Nobody waits for the tail, so it cannot deadlock whatever the dispatch order; a grid
// barrier (all wait for all) was measured at 70-90 us for 600 workgroups, this costs the tail ~1-2 us.
__device__ __forceinline__ void continuity_tail(Clock *clk, int q, const Phys &ph, const FluidTmp &t, int nb)
{
    if (!clk->run[q]) {
        if (threadIdx.x == 0) clk->run[1 - q] = 0;  // idle slot keeps the following slots idle
        return;
    }
    Clock c0;
    int fl = 0;
    if (threadIdx.x == 0) { c0 = *clk; fl = *t.flags; }
    double m = 0.0, d = 0.0;
    int lost = 0;
    const bool track = t.half_skin >= 0.0;
    for (int k = threadIdx.x; k < nb; k += kBlock) {
        if (track) d = fmax(d, t.dpart[k]);  // written by pass CD, a kernel ago
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(&t.vpart[k]);
        unsigned long long bits;
        unsigned spins = 0;
        while ((bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kVpartEmpty) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { lost = 1; break; }  // never hang the device on a protocol bug
        }
        if (!lost) m = fmax(m, __longlong_as_double((long long)bits));
        __hip_atomic_store(slot, kVpartEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    m = wave_max(m);
    d = wave_max(d);
    lost = __any(lost) ? 1 : 0;
    __shared__ double s_m[kBlock / 64], s_d[kBlock / 64];
    __shared__ int s_l[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_d[threadIdx.x >> 6] = d; s_l[threadIdx.x >> 6] = lost; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / 64; ++k) { m = fmax(m, s_m[k]); d = fmax(d, s_d[k]); lost |= s_l[k]; }
        if (lost) c0.status = SPHX_ERR_DIVERGED;
        clock_step(clk, c0, q, ph, sqrt(m), fl, -1, track ? sqrt(d) : -1.0, 0, t.half_skin);
    }
}

// Skinned slabs: the local maxima the step's all-reduce needs -- max |v| and max drift of the owned particles -- are
// reduced by a tail workgroup of pass E (same hand-over as continuity_tail: every vpart entry is its own "ready" flag)
// instead of a one-workgroup kernel behind it.  out[0..1] = {max |v|, max drift}; a stopped loop reports zeros.
__device__ __forceinline__ void slab_seal_tail(const Clock *clk, int q, const FluidTmp &t, int nb)
{
    if (!clk->run[q]) {
        if (threadIdx.x == 0) { t.seal_out[0] = 0.0; t.seal_out[1] = 0.0; }
        return;
    }
    double m = 0.0, d = 0.0;
    int lost = 0;
    for (int k = threadIdx.x; k < nb; k += kBlock) {
        d = fmax(d, t.dpart[k]);  // written by pass CD, a kernel ago
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(&t.vpart[k]);
        unsigned long long bits;
        unsigned spins = 0;
        while ((bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kVpartEmpty) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { lost = 1; break; }  // never hang the device on a protocol bug
        }
        if (!lost) m = fmax(m, __longlong_as_double((long long)bits));
        __hip_atomic_store(slot, kVpartEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    m = wave_max(m);
    d = wave_max(d);
    lost = __any(lost) ? 1 : 0;
    __shared__ double s_m[kBlock / 64], s_d[kBlock / 64];
    __shared__ int s_l[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_d[threadIdx.x >> 6] = d; s_l[threadIdx.x >> 6] = lost; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / 64; ++k) { m = fmax(m, s_m[k]); d = fmax(d, s_d[k]); lost |= s_l[k]; }
        t.seal_out[0] = lost ? INFINITY : sqrt(m);  // (a lost hand-over poisons the maximum: the clock raises DIVERGED)
        t.seal_out[1] = sqrt(d);
    }
}

// tail: 1 = the launch has one workgroup more than the pass needs; it runs continuity_tail (2: slab_seal_tail)
// WALK: the large-channel form of the walk (see the "_w" kernels): entries ahead, fluid / wall loops, fold hoisted
// (bid, nb: this workgroup's index among the nb workgroups of the pass; c_*: the LDS tile arrays of the calling kernel)
// CODED: slot-coded list entries (kSlotCodes)
template <int LPP, bool WALK, int TILE, bool CODED = false>
__device__ __forceinline__ void continuity_body(Clock *clk, int q, const Grid &g, const Phys &ph, const FluidSet &s,
                                                const FluidTmp &t, const Walls &w, int do_hist, int tail, int bid, int nb,
                                                double2 *c_pos, double2 *c_vel, double *c_vol, int next_half = 0)
{
    static_assert(!CODED || (WALK && TILE == kSlotCodes), "slot-coded lists: this pass stages the whole layout");
    const int blk = xcd_block(bid, nb);
    const int tid = blk * kBlock + threadIdx.x;
    const int i = tid / LPP, sub = tid % LPP;
    const bool in_cap = i < t.cap;
    if (WALK && beyond_population<LPP>(clk, t, blk)) {  // nothing here: only the workgroup's entry of the max |v|^2 reduction is owed
        if (threadIdx.x == 0 && clk->run[q] && !next_half) {
            if (tail) __hip_atomic_store(reinterpret_cast<unsigned long long *>(&t.vpart[blk]), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else t.vpart[blk] = 0.0;
        }
        return;
    }
    const double2 pi = in_cap ? s.pos[i] : make_double2(0.0, 0.0);
    const double2 vi = in_cap ? t.veln[i] : make_double2(0.0, 0.0);
    const int packed = t.nl_cnt[tid];
    const int nn_all = list_rows(packed);
    // the first rows are requested together with the count (at 32 lanes per particle a lane rarely owns more than
    // two): count -> entry -> neighbour data becomes {count, entries} -> neighbour data
    // (WALK: the first two words of the packed fluid rows, see FluidTmp::nl_pk)
    const int *first_rows = WALK ? t.nl_pk : t.nl_idx;
    const int e_row0 = first_rows[tid], e_row1 = first_rows[(size_t)t.nl_stride + tid];
    // (rows 2 and 3 only where lanes own that many: few lanes per particle)
    const int e_row2 = (!WALK && LPP <= 8) ? t.nl_idx[2 * (size_t)t.nl_stride + tid] : 0;
    const int e_row3 = (!WALK && LPP <= 8) ? t.nl_idx[3 * (size_t)t.nl_stride + tid] : 0;
    const bool lead = in_cap && sub == 0;
    const double4 a_own = lead ? t.a[i] : make_double4(0.0, 0.0, 0.0, 0.0);
    const double rhoh_i = a_own.z;
    const double2 pn = (lead && do_hist) ? t.posn[i] : make_double2(0.0, 0.0);  // (requested whenever a histogram is possible)
    const TileMap layout = (WALK && TILE > 0) ? tile_map_of(t, blk) : TileMap{0, 0, 0, 0, 0, 0};
    const double dt = clk->dt;
    const bool want_out = !WALK || step_outputs_wanted(clk, t);
    const int cell_own = (g.own_by_cell && lead) ? s.cell[i] : 0;
    if (!clk->run[q]) return;
    const int n_now = clk->n;
    const bool active = i < n_now;
    double rate = 0.0;
    const double xi = pi.x, yi = pi.y, vxi = vi.x, vyi = vi.y;
    // do_hist: 1 = this step re-bins (static schedule); 100 + K = dynamic context: the K-th step since the last re-binning will
    // re-bin whatever the drift says, so its histogram can be taken here (k_bin then skips)
    const bool hist = do_hist == 1 || (do_hist >= 100 && clk->pos_count >= do_hist - 101);
    // Tiled form: the cell histogram goes through LDS.  A particle's new cell lies within a cell or two of the workgroup's old
    // ones, so the workgroup counts into a window of kHistCols x kHistRows cells around its first particle's cell with LDS
    // atomics and adds the window to the global histogram once -- 6 M global atomics, ~60 per cell and step, were 135 us of the
    // pass on every step that takes the histogram (6 M particles).  Anything outside the window is counted globally as before.
    constexpr bool kHistWindow = WALK && TILE > 0;
    constexpr int kHistCols = 3, kHistRows = 48, kHistLead = 4;
    __shared__ int h_win[kHistWindow ? kHistCols * kHistRows : 1];
    int h_cx0 = 0, h_r0 = 0;
    if (kHistWindow && hist) {
        const int c_first = s.cell[min(blk * (kBlock / LPP), max(n_now - 1, 0))];
        h_cx0 = c_first / g.ncy - 1;
        h_r0 = c_first % g.ncy - kHistLead;
        if (threadIdx.x < kHistCols * kHistRows) h_win[threadIdx.x] = 0;  // (the barrier behind the tile staging covers this)
    }
    // The workgroup's entry of the max |v|^2 reduction needs nothing but the particles' own records.  The compact kernels (small
    // channels) publish it HERE, before the walk, so that the tail workgroup reduces and advances the clock while the walks are
    // still running -- the clock update used to start when the last workgroup had finished (C1 14.55 -> 14.25, C2 17.55 -> 17.35
    // us/step).  (Every workgroup has read what it needs of the clock by the time it publishes -- dt, the run flag, the
    // population are read above -- and the tail cannot finish before all of them have published.)  The large-channel forms
    // publish at the end as before: there the extra barrier in front of the walk costs more than the early clock gains (C3
    // 44.5 -> 45.2 us/step with it).
    auto publish_vmax = [&]() {
        double v2 = 0.0;
        if (active && sub == 0 && owns(g, xi, cell_own)) {
            v2 = vxi * vxi + vyi * vyi;
            if (v2 != v2) v2 = INFINITY;  // NaN poisons the max on purpose
        }
        v2 = wave_max(v2);
        __shared__ double s_max[kBlock / 64];
        if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = v2;
        __syncthreads();
        if (threadIdx.x == 0 && !next_half) {  // (an inner sub-step leaves the "ready" slots of the step's last pass E alone)
            double m = s_max[0];
            for (int k = 1; k < kBlock / 64; ++k) m = fmax(m, s_max[k]);
            if (tail)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(&t.vpart[blk]), (unsigned long long)__double_as_longlong(m),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                t.vpart[blk] = m;
        }
    };
    if (!WALK) publish_vmax();
    if (WALK) {
        const int rows = active ? nn_all : 0, rows_fl = active ? list_fluid_rows(packed) : 0;
        TileMap tm{0, 0, 0, 0, 0, 0};
        if (TILE > 0) {
            tm = staged_map<LPP>(layout, blk, n_now, TILE);
            static_assert(TILE <= 2 * kBlock, "two slots per thread");
            {  // (both of a thread's slots requested before either is waited for: see k_forces_w)
                const int total = tm.total(), sl0 = threadIdx.x, sl1 = threadIdx.x + kBlock;
                const bool h0 = sl0 < total, h1 = sl1 < total;
                const int k0 = tm.index(h0 ? sl0 : 0) + n2_prologue_cost(sl0, tm), k1 = tm.index(h1 ? sl1 : 0) + n2_prologue_cost(sl1, tm);
                double2 p0, p1, v0, v1;
                double u0, u1;
                if (h0) { p0 = s.pos[k0]; v0 = t.veln[k0]; u0 = t.vol[k0]; }
                if (h1) { p1 = s.pos[k1]; v1 = t.veln[k1]; u1 = t.vol[k1]; }
                if (h0) { c_pos[sl0] = p0; c_vel[sl0] = v0; c_vol[sl0] = u0; }
                if (h1) { c_pos[sl1] = p1; c_vel[sl1] = v1; c_vol[sl1] = u1; }
            }
            __syncthreads();
        }
        auto fetch = [&](int k, double2 &pj, double2 &vj, double &Volj) {
            if (TILE > 0) {  // (early return, not if/else: see k_kgc_w)
                const int slot = tm.slot(k);
                if (slot >= 0) { pj = lds_double2(c_pos, slot); vj = lds_double2(c_vel, slot); Volj = lds_double(c_vol, slot); return; }
            }
            pj = s.pos[k]; vj = t.veln[k]; Volj = t.vol[k];
        };
        auto term = [&](double dx, double dy, double ujx, double ujy, double Volj) {
            const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
            const double ex = dx * inv_r, ey = dy * inv_r;
            rate += ((vxi - ujx) * ex + (vyi - ujy) * ey) * spline_dW_sel(ph.kc, r) * Volj;
        };
        auto fetch_code = [&](int e, double2 &pj, double2 &vj, double &Volj) {  // (CODED)
#if SPHX_EXP_PRETEND_COMPLETE_TILE & 8
            e = min(e, kSlotCodes - 1);  // MEASUREMENT ONLY, see k_forces_w
#endif
            if (e < kSlotCodes) { pj = lds_double2(c_pos, e); vj = lds_double2(c_vel, e); Volj = lds_double(c_vol, e); return; }
            const int k = wrap_index(i + e - kCodeBias, n_now);
            pj = s.pos[k]; vj = t.veln[k]; Volj = t.vol[k];
        };
        const bool seam = __any(active && near_seam(g, xi));
        const bool all_near = CODED && !__any(packed & kFarTileBit);  // (see kFarTileBit)
        if (CODED && seam)
            walk_fluid_rows<true>(t, tid, i, rows_fl, e_row0, e_row1, [&](int e) {
                double2 pj, vj; double Volj;
                fetch_code(e, pj, vj, Volj);
                term(min_image(g, xi - pj.x), yi - pj.y, vj.x, vj.y, Volj);
            });
        else if (CODED && all_near)
            walk_fluid_rows<true>(t, tid, i, rows_fl, e_row0, e_row1, [&](int e) {
                const double2 pj = lds_double2(c_pos, e), vj = lds_double2(c_vel, e);
                term(xi - pj.x, yi - pj.y, vj.x, vj.y, lds_double(c_vol, e));
            });
        else if (CODED)
            walk_fluid_rows<true>(t, tid, i, rows_fl, e_row0, e_row1, [&](int e) {
                double2 pj, vj; double Volj;
                fetch_code(e, pj, vj, Volj);
                term(xi - pj.x, yi - pj.y, vj.x, vj.y, Volj);
            });
        else if (seam)
            walk_fluid_rows(t, tid, i, rows_fl, e_row0, e_row1, [&](int k) {
                double2 pj, vj; double Volj;
                fetch(wrap_index(k, n_now), pj, vj, Volj);
                term(min_image(g, xi - pj.x), yi - pj.y, vj.x, vj.y, Volj);
            });
        else
            walk_fluid_rows(t, tid, i, rows_fl, e_row0, e_row1, [&](int k) {
                double2 pj, vj; double Volj;
                fetch(k, pj, vj, Volj);
                term(xi - pj.x, yi - pj.y, vj.x, vj.y, Volj);
            });
        walk_wall_rows(t, tid, rows_fl, rows, [&](int k) {
            const double2 pj = w.pos[k];
            const double4 wj = w.a[k];  // {Vol, vx, vy, -}; the wall presents its velocity mirrored through the particle's
            term(min_image(g, xi - pj.x), yi - pj.y, 2.0 * wj.y - vxi, 2.0 * wj.z - vyi, wj.x);
        });
    } else if (active) {
        for (int m = 0; m < nn_all; ++m) {
            const int e = m == 0 ? e_row0 : (m == 1 ? e_row1 : (LPP <= 8 && m == 2 ? e_row2 : (LPP <= 8 && m == 3 ? e_row3 : t.nl_idx[(size_t)m * t.nl_stride + tid])));
            const bool wall = (e & kWallBit) != 0;
            const int k = e & (kWallBit - 1);
            const double2 pj = (wall ? w.pos : (const double2 *)s.pos)[k];
            double2 vj;
            double Volj;
            if (wall) {
                const double4 wj = w.a[k];  // {Vol, vx, vy, -}
                Volj = wj.x;
                vj = make_double2(2.0 * wj.y - vxi, 2.0 * wj.z - vyi);  // mirrored wall velocity
            } else {
                Volj = t.vol[k];
                vj = t.veln[k];
            }
            const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
            const double r2 = dx * dx + dy * dy, inv_r = rsqrt_nr(r2), r = r2 * inv_r;
            const double ex = dx * inv_r, ey = dy * inv_r;
            const double u_jump = (vxi - vj.x) * ex + (vyi - vj.y) * ey;
            rate += u_jump * spline_dW_sel(ph.kc, r) * Volj;
        }
    }
    rate = group_sum<LPP>(rate);
    if (active && sub == 0) {
        const double rhoh = rhoh_i;
        const double drho_new = rate * rhoh;
        double rho = rhoh + drho_new * (0.5 * dt);
        if (rho < 1e-10) rho = ph.rho0;
        t.drhon[i] = drho_new;
        if (want_out) {
            t.rho_out[i] = rho;
            t.p_out[i] = eos_pressure(rho, ph.rho0, ph.p0);
        }
        if (next_half) {  // dual-rate loop: another sub-step follows, with the density carried on (not re-summed)
            double rhoh2, p2;
            half_state(ph, rho, drho_new, dt, rhoh2, p2);
            t.a[i] = make_double4(a_own.x, p2, rhoh2, rho);
        }
        if (hist) {
            int cx, cy;
            cell_of(g, pn.x, pn.y, cx, cy);
            const int c = cx * g.ncy + cy;
            t.cellid[i] = c;
            bool counted = false;
            if (kHistWindow) {
                int dcol = cx - h_cx0;
                if (g.periodic) dcol = dcol < 0 ? dcol + g.ncx : (dcol >= g.ncx ? dcol - g.ncx : dcol);
                const int drow = cy - h_r0;
                if ((unsigned)dcol < (unsigned)kHistCols && (unsigned)drow < (unsigned)kHistRows) {
                    atomicAdd(&h_win[dcol * kHistRows + drow], 1);
                    counted = true;
                }
            }
            if (!counted) atomicAdd(&t.count[c], 1);
        }
    }
    if (WALK) publish_vmax();  // (its barrier also closes the window histogram)
    if (kHistWindow && hist && threadIdx.x < kHistCols * kHistRows) {
        const int v = h_win[threadIdx.x];
        if (v) {
            int col = h_cx0 + (int)threadIdx.x / kHistRows;
            if (g.periodic) col = col < 0 ? col + g.ncx : (col >= g.ncx ? col - g.ncx : col);
            const int row = h_r0 + (int)threadIdx.x % kHistRows;
            atomicAdd(&t.count[col * g.ncy + row], v);  // (v > 0: somebody's cell, so col and row are inside the grid)
        }
    }
}

// (waves_per_eu: the large-channel forms of passes E and A fit eight waves per SIMD by their vector registers (62-67) but took 106
//  scalar registers -- seven waves; they are latency-bound, the eighth wave is worth 9 % of pass E at 6 M particles.  Asked for,
//  the compiler finds a 78-SGPR allocation without spills.)
template <int LPP, bool WALK, int TILE, bool CODED = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WALK ? 8 : 1))) void k_continuity(Clock *clk, int q, Grid g, Phys ph,
                                                       FluidSet s, FluidTmp t, Walls w, int do_hist, int tail, int next_half)
{
    constexpr int kSlots = TILE > 0 ? TILE : 1;
    __shared__ double2 c_pos[kSlots], c_vel[kSlots];
    __shared__ double c_vol[kSlots];
    const int nb = (int)gridDim.x - (tail ? 1 : 0);
    if (tail && (int)blockIdx.x == nb) {
        if (tail == 2) slab_seal_tail(clk, q, t, nb);
        else continuity_tail(clk, q, ph, t, nb);
        return;
    }
    continuity_body<LPP, WALK, TILE, CODED>(clk, q, g, ph, s, t, w, do_hist, tail, (int)blockIdx.x, nb, c_pos, c_vel, c_vol, next_half);
}

// Small channels, steps that do not re-bin: pass E of this step and pass A of the NEXT step in one launch, side by
// side.  Both depend only on pass CD (E on the kicked velocities, A on the new positions) and not on each other, and at
// a few thousand particles the chip is mostly idle, so the launch takes as long as the longer of the two instead of
// their sum plus a kernel boundary.  Workgroups [0, nb): pass E on state s with the step's list (t); [nb, 2 nb): pass A
// on the new state s_next, walking the superset list, into the other list / record buffers (t_next) -- without the
// half-step density and pressure, which need the next step's dt (the tail workgroup of this very launch computes it):
// pass B of the next step closes that (k_kgc, finish_half); workgroup 2 nb: the clock (continuity_tail).
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_continuity_density(Clock *clk, int q, Grid g, Phys ph, FluidSet s, FluidTmp t,
                                                               Walls w, FluidSet s_next, FluidTmp t_next, int with_tail)
{
    const int nb = ((int)gridDim.x - with_tail) / 2;  // (with_tail = 0: kernel timing, the clock must not advance)
    const int b = (int)blockIdx.x;
    if (with_tail && b == 2 * nb) {
        continuity_tail(clk, q, ph, t, nb);
        return;
    }
    if (b < nb) continuity_body<LPP, false, 0>(clk, q, g, ph, s, t, w, 0, with_tail, b, nb, nullptr, nullptr, nullptr);
    else density_body<LPP, 2>(clk, q, g, ph, s_next, t_next, w, b - nb, nb, false);
}

// the same with the large-channel forms of the two passes (mid-size channels: 4-8 lanes per particle, clock in the tail)
template <int LPP>
__global__ __launch_bounds__(kBlock) void k_continuity_density_w(Clock *clk, int q, Grid g, Phys ph, FluidSet s, FluidTmp t,
                                                                 Walls w, FluidSet s_next, FluidTmp t_next, int with_tail)
{
    const int nb = ((int)gridDim.x - with_tail) / 2;  // (with_tail = 0: kernel timing, the clock must not advance)
    const int b = (int)blockIdx.x;
    if (with_tail && b == 2 * nb) {
        continuity_tail(clk, q, ph, t, nb);
        return;
    }
    if (b < nb) continuity_body<LPP, true, 0>(clk, q, g, ph, s, t, w, 0, with_tail, b, nb, nullptr, nullptr, nullptr);
    else density_walk_body<LPP>(clk, q, g, ph, s_next, t_next, w, b - nb, nb, false);
}

// standalone cell histogram (context creation, wall grid, slab steps): same binning as pass E
__global__ __launch_bounds__(kBlock) void k_bin(const Clock *clk, int q, Grid g, int n_fixed, const double2 *pos,
                                                int *cellid, int *count)
{
    if (clk && !slot_active(clk, q)) return;
    const int n = clk ? clk->n : n_fixed;
    // grid-stride: dynamic contexts launch the re-binning kernels on every step with a small grid, so that the
    // launches that skip cost next to nothing
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int cx, cy;
        const double2 p = pos[i];
        cell_of(g, p.x, p.y, cx, cy);
        const int c = cx * g.ncy + cy;
        cellid[i] = c;
        atomicAdd(&count[c], 1);
    }
}

// block-wide exclusive scan of one int per thread (kScanBlock threads); returns the block total
__device__ __forceinline__ int block_exclusive_scan(int v, int &total, int *s_wave /*[17]*/)
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
        const int ws = lane < kScanBlock / 64 ? s_wave[lane] : 0;
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

// exclusive scan of count[0..n) into start[0..n], by one block
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

// big grids: tile-local exclusive scan (one 1024-cell tile per block) + tile sums
__global__ __launch_bounds__(kScanBlock) void k_scan_tiles(const Clock *clk, int q, const int *count, int *start,
                                                           int *tile_sum, int n)
{
    if (clk && !slot_active(clk, q)) return;
    __shared__ int s_wave[kScanBlock / 64 + 1];
    const int idx = blockIdx.x * kScanBlock + (int)threadIdx.x;
    const int v = idx < n ? count[idx] : 0;
    int total;
    const int ex = block_exclusive_scan(v, total, s_wave);
    if (idx < n) start[idx] = ex;
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add(const Clock *clk, int q, int *start, const int *tile_off,
                                                         int n, int n_tiles)
{
    if (clk && !slot_active(clk, q)) return;
    const int idx = blockIdx.x * kScanBlock + (int)threadIdx.x;
    if (idx < n) start[idx] += tile_off[blockIdx.x];
    if (idx == 0) start[n] = tile_off[n_tiles];
}

// Millions of particles leave ~10^5 per-block maxima; one workgroup reading them all took 84 us at 6 M particles.
// This pre-pass folds kMaxTile of them per workgroup (max is exact, so the result does not depend on the split).
constexpr int kMaxTile = 4096;
__global__ __launch_bounds__(kScanBlock) void k_max_tiles(const Clock *clk, int q, int n, const double *a, const double *b,
                                                          double *out_a, double *out_b)
{
    if (!clk->run[q]) return;
    const int base = blockIdx.x * kMaxTile;
    double m = 0.0, d = 0.0;
    for (int k = base + threadIdx.x; k < min(base + kMaxTile, n); k += kScanBlock) {
        m = fmax(m, a[k]);
        if (b) d = fmax(d, b[k]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m = fmax(m, __shfl_xor(m, off)); d = fmax(d, __shfl_xor(d, off)); }
    __shared__ double s_m[kScanBlock / 64], s_d[kScanBlock / 64];
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_d[threadIdx.x >> 6] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kScanBlock / 64; ++k) { m = fmax(m, s_m[k]); d = fmax(d, s_d[k]); }
        out_a[blockIdx.x] = m;
        if (b) out_b[blockIdx.x] = d;
    }
}

// Step kernel 5: finish the clock of this step (vmax -> next dt, t += dt, stop test) and scan the
// cell histogram (small grids) or the tile sums (big grids).  Single block.
//   vmax_global: slab mode -- the all-reduced max |v| replaces the local reduction.
//   dpart / rebuilt / half_skin: displacement bookkeeping of grids that are rebuilt only every few steps.
//   slab_counters: slab mode -- the keep/left/right counters of k_slab_pack, zeroed for the next step (the unpack
//   kernel of this step has read them).  vpart_reset: see continuity_tail.
__global__ __launch_bounds__(kScanBlock) void k_clock_scan(Clock *clk, int q, Phys ph, int n_vpart,
                                                           const double *vpart, const double *vmax_global,
                                                           const int *flags, const int *count,
                                                           int *start_next, int n_scan, const int *n_new,
                                                           const double *dpart, int rebuilt, double half_skin,
                                                           int *slab_counters, unsigned long long *vpart_reset, int dyn_K)
{
    // everything is requested before the run flag is looked at (stale values are harmless when the slot turns
    // out to be idle); only the thread that advances the clock loads it
    Clock c0;
    int fl = 0, nn = -1;
    double vg = 0.0;
    if (threadIdx.x == 0) {
        c0 = *clk;
        fl = *flags;
        if (n_new) nn = *n_new;
        if (vmax_global) vg = *vmax_global;
    }
    double m = 0.0, d = 0.0;
    if (!vmax_global)
        for (int k = threadIdx.x; k < n_vpart; k += kScanBlock) m = fmax(m, vpart[k]);
    if (dpart)
        for (int k = threadIdx.x; k < n_vpart; k += kScanBlock) d = fmax(d, dpart[k]);
    if (!clk->run[q]) {
        if (threadIdx.x == 0) clk->run[1 - q] = 0;
        return;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { m = fmax(m, __shfl_xor(m, off)); d = fmax(d, __shfl_xor(d, off)); }
    __shared__ double s_m[kScanBlock / 64], s_d[kScanBlock / 64];
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_d[threadIdx.x >> 6] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kScanBlock / 64; ++k) { m = fmax(m, s_m[k]); d = fmax(d, s_d[k]); }
        // max of sqrt == sqrt of max (monotone, correctly rounded)
        clock_step(clk, c0, q, ph, vmax_global ? vg : sqrt(m), fl, nn, dpart ? sqrt(d) : -1.0, rebuilt, half_skin, dyn_K);
        if (slab_counters) { slab_counters[0] = 0; slab_counters[1] = 0; slab_counters[2] = 0; }  // pack counters of the next step
    }
    // contexts whose move steps use the tail workgroup of pass E expect "empty" entries before every pass E
    if (vpart_reset)
        for (int k = threadIdx.x; k < n_vpart; k += kScanBlock) vpart_reset[k] = kVpartEmpty;
    if (count) scan_counts(count, start_next, n_scan);
}

__global__ __launch_bounds__(kScanBlock) void k_scan_only(const Clock *clk, int q, const int *count, int *start, int n)
{
    if (clk && !slot_active(clk, q)) return;
    scan_counts(count, start, n);
}

// Step kernel 6: place every particle index into its cell range (arrival order, made canonical by
// k_reorder).  atomicSub counts the histogram back down to zero, ready for the next step.
__global__ __launch_bounds__(kBlock) void k_scatter(const Clock *clk, int q, int n_fixed, const int *cellid,
                                                    int *count, const int *start_next, int *perm)
{
    if (clk && !slot_active(clk, q)) return;
    const int n = clk ? clk->n : n_fixed;
    // Particles arrive nearly sorted (the previous cell order), so the 64 of a wavefront fall into a handful of cells:
    // one returning atomic per (wavefront, cell) instead of one per particle -- the lanes of a cell take consecutive slots
    // of the range their leader reserved.  (6 M returning atomics were 330 us; the order inside a cell is made canonical
    // by k_reorder anyway.)
    const int lane = threadIdx.x & 63;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {  // (uniform trip count per workgroup)
        const int i = base + (int)threadIdx.x;
        const bool live = i < n;
        const int c = live ? cellid[i] : -1;
        unsigned long long todo = __ballot(live);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int c_lead = __shfl(c, lead);
            const unsigned long long same = __ballot(live && c == c_lead) & todo;
            if (live && c == c_lead && (todo >> lane & 1ull)) {
                const int cnt = __popcll(same), rank = __popcll(same & below);
                int first = 0;
                if (rank == 0) first = atomicSub(&count[c], cnt);  // returns the count before: slots first-cnt .. first-1
                first = __shfl(first, __ffsll((long long)same) - 1);
                perm[start_next[c] + first - 1 - rank] = i;
            }
            todo &= ~same;
        }
    }
}

struct ReorderArgs {
    int n1, n2, n4;  // number of 8-, 16- and 32-byte fields
    const double *src1[2];
    double *dst1[2];
    const double2 *src2[3];
    double2 *dst2[3];
    const double4 *src4[1];
    double4 *dst4[1];
    const int *id_src;
    int *id_dst;
    int *src_of;
    int *cell_dst;  // binned cell of every destination slot (nullptr: not needed, walls)
    int *slot_of_id;  // skinned slabs: particle id -> its slot in the new ordering (nullptr: not needed)
};

// Step kernel 7: canonical rank inside the cell (ascending particle id -> an order that does not
// depend on arrival order, launch shape or domain decomposition) and the gather of every persistent
// field into the new ordering.
__global__ __launch_bounds__(kBlock) void k_reorder(const Clock *clk, int q, int n_fixed, const int *cellid,
                                                    const int *start_next, const int *perm, ReorderArgs a)
{
    if (clk && !slot_active(clk, q)) return;
    const int n = clk ? clk->n : n_fixed;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int c = cellid[i];
        const int lo = start_next[c], hi = start_next[c + 1];
        const int my_id = a.id_src[i];
        int rank = 0;
        for (int k = lo; k < hi; ++k) {
            const int o = perm[k];
            const int oid = a.id_src[o];
            rank += (oid < my_id || (oid == my_id && o < i)) ? 1 : 0;  // tie (periodic images in a slab): by slot
        }
        const int dst = lo + rank;
#pragma unroll
        for (int f = 0; f < 3; ++f)
            if (f < a.n2) a.dst2[f][dst] = a.src2[f][i];
#pragma unroll
        for (int f = 0; f < 2; ++f)
            if (f < a.n1) a.dst1[f][dst] = a.src1[f][i];
        if (a.n4) a.dst4[0][dst] = a.src4[0][i];
        a.id_dst[dst] = my_id;
        if (a.src_of) a.src_of[dst] = i;
        if (a.cell_dst) a.cell_dst[dst] = c;
        if (a.slot_of_id) a.slot_of_id[my_id] = dst;
    }
}

// both step-slot flags off: whatever is enqueued next returns at once (sphx_ctx_prepare_steps warms graphs this way)
__global__ void k_disarm(Clock *clk)
{
    clk->run[0] = 0;
    clk->run[1] = 0;
}

__global__ void k_rebinned(Clock *clk)
{
    clk->drift = 0.0;
    clk->need_rebuild = 0;
}

__global__ __launch_bounds__(kBlock) void k_iota(int n, int *a, int base)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) a[i] = base + i;
}

__global__ __launch_bounds__(kBlock) void k_wrap_x(int n, double2 *pos, double DL)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) pos[i].x = wrap_x(pos[i].x, DL);
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

// initial max |v| over owned particles (vecnorm over the fluid, SPH_Poiseuille.m:521); single block
__global__ __launch_bounds__(kScanBlock) void k_vmax_init(Clock *clk, Grid g, const double2 *pos, const double2 *vel,
                                                          double *vmax_out, const int *cell = nullptr)
{
    const int n = clk->n;
    double m = 0.0;
    for (int k = threadIdx.x; k < n; k += kScanBlock) {
        const double x = pos[k].x;
        if (!owns(g, x, (g.own_by_cell && cell) ? cell[k] : 0)) continue;
        const double2 v = vel[k];
        double v2 = v.x * v.x + v.y * v.y;
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
        if (vmax_out) *vmax_out = sqrt(m);
    }
}

// ---------------------------------------------------------------------------------------------
// monitors and pair-list emission on the current ordering
// ---------------------------------------------------------------------------------------------

// wall shear (sph_physics_mex.c:1713-1742): new neighbour structure, new pos/vel, Vol/B of the step
// that just finished (reached through src_of).  Per-block partial sums, reduced by k_tau_final.
__global__ __launch_bounds__(kBlock) void k_wall_shear(const Clock *clk, Grid g, Phys ph, FluidSet s, FluidTmp t,
                                                       Walls w, int use_src, double *part /*[2*grid]*/)
{
    const int n = clk->n;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double fb = 0.0, ft = 0.0;
    if (i < n) {
        const double xi = s.pos[i].x, yi = s.pos[i].y;
        int cx, cy;
        binned_cell(g, s, i, cx, cy);
        if (w.row_any[cy] && xi >= g.own_lo && xi < g.own_hi) {
            const int o = use_src ? t.src_of[i] : i;
            const double Voli = t.a[o].x;
            const double4 Bo = t.B[o];
            const double b11 = Bo.x, b12 = Bo.y, b21 = Bo.z, b22 = Bo.w;
            const double vxi = s.vel[i].x;
            sweep<1>(g, w.start, cx, cy, 0, [&](int k) {
                const double2 pj = w.pos[k];
                const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
                const double r2 = dx * dx + dy * dy;
                if (r2 > kR2Min && r2 < ph.kc.rcut2) {
                    const double4 wj = w.a[k];
                    const double r = sqrt(r2);
                    const double ex = dx / r, ey = dy / r;
                    const double eBe = ex * (b11 * ex + b12 * ey) + ey * (b21 * ex + b22 * ey);
                    const double f = 4.0 * ph.mu * eBe * spline_dW(ph.kc, r) * wj.x * (vxi - wj.y) /
                                     (r + 0.01 * ph.kc.h) * Voli;
                    const double yj = pj.y;
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
        out[0] = -a / DL;  // the caller sums slabs; -sum/DL is linear
        out[1] = -b / DL;
    }
}

// Pair emission in the MEX convention (sph_neighbor_search_mex.c:353-383): a fluid-fluid pair is
// produced once, from the particle with the smaller ORIGINAL index; fluid-wall pairs always.
// MODE 0: count into cnt[orig]; MODE 1: write at off[orig].  (single-GPU contexts only)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_pairs(const Clock *clk, Grid g, Phys ph, FluidSet s, Walls w, int *cnt,
                                                  const int *off, double *o_i, double *o_j, double *o_dx,
                                                  double *o_dy, double *o_r, double *o_W, double *o_dW)
{
    const int nf = clk->n;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nf) return;
    const double xi = s.pos[i].x, yi = s.pos[i].y;
    const int a = s.id[i];
    int cx, cy;
    binned_cell(g, s, i, cx, cy);
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
    sweep<1>(g, s.start, cx, cy, 0, [&](int k) {
        const double2 pj = s.pos[k];
        const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
        const double r2 = dx * dx + dy * dy;
        if (r2 > kR2Min && r2 < ph.kc.rcut2) {
            const int b = s.id[k];
            if (b > a) emit(b, dx, dy, r2);
        }
    });
    if (w.row_any[cy]) {
        sweep<1>(g, w.start, cx, cy, 0, [&](int k) {
            const double2 pj = w.pos[k];
            const double dx = min_image(g, xi - pj.x), dy = yi - pj.y;
            const double r2 = dx * dx + dy * dy;
            if (r2 > kR2Min && r2 < ph.kc.rcut2) emit(w.id[k], dx, dy, r2);
        });
    }
    if (!MODE) cnt[a] = n;
}

// scatter one component of a sorted field (records of `stride` doubles) back to the caller's row numbering;
// src_of (optional): the field is stored in the ordering before the last re-binning, slot i was slot src_of[i] there
__global__ __launch_bounds__(kBlock) void k_unsort(int n, const int *id, const double *src, int stride, double *dst,
                                                   const int *src_of)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[id[i]] = src[(size_t)(src_of ? src_of[i] : i) * stride];
}

// Dynamic contexts re-bin in place: k_reorder gathers into temporaries, this copies them back over the state of
// the next step and the layout arrays (only on steps that re-bin).
struct CopyBack {
    const double2 *pos_s, *vel_s, *posb_s;
    double2 *pos_d, *vel_d, *posb_d;
    const double *drho_s, *mass_s;
    double *drho_d, *mass_d;
    const int *id_s, *cell_s, *start_s;
    int *id_d, *cell_d, *start_d;
    int n_start;  // ncells + 1
    // 1: binning positions, cells and cell starts were written in place -- nobody reads the old ones during the re-ordering, so
    // k_scan_* and k_reorder can take the layout's own arrays as their destination -- and only what the re-ordering itself reads
    // while it writes (state, mass, id) went through temporaries: 52 bytes per particle to bring back instead of 72
    int lean;
};
__global__ __launch_bounds__(kBlock) void k_copyback(const Clock *clk, int qf, CopyBack a)
{
    if (!slot_active(clk, qf)) return;
    const int n = clk->n;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < (a.lean ? n : max(n, a.n_start)); i += gridDim.x * kBlock) {
        if (i < n) {
            a.pos_d[i] = a.pos_s[i];
            a.vel_d[i] = a.vel_s[i];
            a.drho_d[i] = a.drho_s[i];
            a.mass_d[i] = a.mass_s[i];
            a.id_d[i] = a.id_s[i];
            if (!a.lean) {
                a.posb_d[i] = a.posb_s[i];
                a.cell_d[i] = a.cell_s[i];
            }
        }
        if (!a.lean && i < a.n_start) a.start_d[i] = a.start_s[i];
    }
}

__global__ __launch_bounds__(kBlock) void k_fill(int n, double *dst, double v)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = v;
}

// ---------------------------------------------------------------------------------------------
// x-slab halo exchange (multi-GPU): pack the end-of-step state of OWNED particles into
//   keep  : still inside this rank's window [win_lo, win_hi) -> compacted for the next step (a particle
//           that crossed into a neighbour's columns stays here as a halo copy: its new owner did not
//           own it during this step and therefore does not send it back yet)
//   sendL : x_new < own_lo + halo_w  (shifted by shift_l: +DL on the first slab)
//   sendR : x_new >= own_hi - halo_w (shifted by shift_r: -DL on the last slab)
// A message is double[1 + 7*cap]: count, then x,y,vx,vy,drho,mass,id blocks of `cap`.
// Particles that were halo copies at the start of the step are dropped: their owner sends fresh ones.
// ---------------------------------------------------------------------------------------------
struct SlabPack {
    double *send_l, *send_r;   // device message buffers
    int *counters;             // [3]: keep, left, right (zeroed by k_slab_unpack of the previous step)
    double2 *kpos, *kvel;      // keep arrays (compacted)
    double *kdrho, *kmass;
    int *kid;
    int *cellid, *count;       // cell of every kept / received particle and the cell histogram (binned on the fly)
    double halo_w, shift_l, shift_r, win_lo, win_hi;
    int msg_cap, keep_cap;
};

__device__ __forceinline__ void msg_put(double *msg, int cap, int slot, double x, double y, double vx, double vy,
                                        double drho, double mass, int id)
{
    double *b = msg + 1;
    b[slot] = x;
    b[(size_t)cap + slot] = y;
    b[2 * (size_t)cap + slot] = vx;
    b[3 * (size_t)cap + slot] = vy;
    b[4 * (size_t)cap + slot] = drho;
    b[5 * (size_t)cap + slot] = mass;
    b[6 * (size_t)cap + slot] = (double)id;
}

__global__ __launch_bounds__(kBlock) void k_slab_pack(const Clock *clk, int q, Grid g, FluidSet s, FluidTmp t,
                                                      SlabPack p)
{
    if (!clk->run[q]) return;
    const int n = clk->n;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double x_ref = s.pos[i].x;
    if (!(x_ref >= g.own_lo && x_ref < g.own_hi)) return;  // halo copy
    const double2 pn = t.posn[i], vn = t.veln[i];
    const double xn = pn.x, yn = pn.y, vxn = vn.x, vyn = vn.y, dr = t.drhon[i], m = s.mass[i];
    const int id = s.id[i];
    if (xn >= p.win_lo && xn < p.win_hi) {
        const int k = atomicAdd(&p.counters[0], 1);
        if (k < p.keep_cap) {
            p.kpos[k] = pn; p.kvel[k] = vn; p.kdrho[k] = dr; p.kmass[k] = m; p.kid[k] = id;
            int cx, cy;
            cell_of(g, xn, yn, cx, cy);
            const int c = cx * g.ncy + cy;
            p.cellid[k] = c;
            atomicAdd(&p.count[c], 1);
        } else atomicOr(t.flags, 2);
    }
    if (xn < g.own_lo + p.halo_w) {
        const int k = atomicAdd(&p.counters[1], 1);
        if (k < p.msg_cap) msg_put(p.send_l, p.msg_cap, k, xn + p.shift_l, yn, vxn, vyn, dr, m, id);
        else atomicOr(t.flags, 2);
    }
    if (xn >= g.own_hi - p.halo_w) {
        const int k = atomicAdd(&p.counters[2], 1);
        if (k < p.msg_cap) msg_put(p.send_r, p.msg_cap, k, xn + p.shift_r, yn, vxn, vyn, dr, m, id);
        else atomicOr(t.flags, 2);
    }
}

// after the pack kernel, one workgroup: publish the message counts and reduce the per-block max |v|^2 of the
// owned particles to max |v| (input of the all-reduce)
__global__ __launch_bounds__(kScanBlock) void k_slab_seal_vmax(const Clock *clk, int q, SlabPack p, int n_vpart,
                                                               const double *vpart, double *vmax_out)
{
    const bool run = clk->run[q] != 0;
    if (threadIdx.x == 0) {
        p.send_l[0] = run ? (double)min(p.counters[1], p.msg_cap) : -1.0;
        p.send_r[0] = run ? (double)min(p.counters[2], p.msg_cap) : -1.0;
    }
    if (!run) { if (threadIdx.x == 0) *vmax_out = 0.0; return; }
    double m = 0.0;
    for (int k = threadIdx.x; k < n_vpart; k += kScanBlock) m = fmax(m, vpart[k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    __shared__ double s_m[kScanBlock / 64];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kScanBlock / 64; ++k) m = fmax(m, s_m[k]);
        *vmax_out = sqrt(m);
    }
}

// append the received halo/migrant particles behind the kept ones; writes the new particle count.
__global__ __launch_bounds__(kBlock) void k_slab_unpack(const Clock *clk, int q, Grid g, SlabPack p, const double *recv_l,
                                                        const double *recv_r, int *n_new, int *flags)
{
    if (!clk->run[q]) return;
    const int nk = min(p.counters[0], p.keep_cap);
    const int nl = (int)recv_l[0], nr = (int)recv_r[0];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (nl < 0 || nr < 0 || nk + nl + nr > p.keep_cap) {
        if (i == 0) { atomicOr(flags, 2); *n_new = min(nk, p.keep_cap); }
        return;
    }
    if (i == 0) *n_new = nk + nl + nr;
    const int cap = p.msg_cap;
    if (i < nl + nr) {
        const double *b = (i < nl ? recv_l : recv_r) + 1;
        const int sl = i < nl ? i : i - nl;
        const int d = nk + i;
        const double x = b[sl], y = b[(size_t)cap + sl];
        p.kpos[d] = make_double2(x, y);
        int cx, cy;
        cell_of(g, x, y, cx, cy);
        const int c = cx * g.ncy + cy;
        p.cellid[d] = c;
        atomicAdd(&p.count[c], 1);
        p.kvel[d] = make_double2(b[2 * (size_t)cap + sl], b[3 * (size_t)cap + sl]);
        p.kdrho[d] = b[4 * (size_t)cap + sl];
        p.kmass[d] = b[5 * (size_t)cap + sl];
        p.kid[d] = (int)b[6 * (size_t)cap + sl];
    }
}

// ---------------------------------------------------------------------------------------------
// Skinned slabs: a slab re-bins (and re-negotiates its halo) only on the K-th step or when the drift bound is hit -- the
// device decides, from ALL-REDUCED max |v| and max drift, so every rank takes the same decision without the host.  In
// between the layout is frozen: each rank sends the new state of a fixed list of boundary particles (send_idx) and the
// receiver writes it into fixed slots (recv_slot); the lists are rebuilt with the layout.  Per step:
//   passes A..E (the local maxima come out of pass E's tail workgroup, slab_seal_tail) -> [max all-reduce of {max|v|, max drift}]
//   -> k_slab_pack3 (decision, message A, clock) -> [to both ring neighbours: message A, and message B = the ids of the send
//   lists made in the PREVIOUS step] -> k_slab_unpack3 (new ids -> slots; message A) -> {re-binning chain, k_slab_sendlist:
//   only when the clock says so}
// (round 3: ten launches of this sequence folded into three -- "last workgroup out" epilogues instead of one-thread kernels)
// Ownership goes by the binned column (Grid::own_by_cell).  A particle that crosses a slab boundary changes owner at
// the next re-binning: both ranks hold it (the old owner keeps everything in its window), the new owner names it in
// its id list and the old owner finds its copy through slot_of_id.
// ---------------------------------------------------------------------------------------------
struct SlabLists {
    int *send_idx[2];   // [msg_cap] slots whose state goes to the left / right neighbour every step
    int *send_cnt;      // [2]
    int *recv_slot[2];  // [msg_cap] slots the entries of the left / right neighbour's messages are written to
    int *recv_cnt;      // [2]
    int *slot_of_id;    // [global fluid count] particle id -> slot in the current layout (stale once a particle has left)
    int *ids_send[2];   // message B, outgoing: [1 + msg_cap] count, ids (device buffers owned by the context)
};

// "Last workgroup out": every workgroup of a kernel takes a ticket when its work is done; the one that draws the last
// ticket sees the complete result of all the others (counters are read with agent-scope atomics) and finishes the
// kernel's single-thread epilogue -- what used to be a one-thread kernel of its own behind it.  Returns true in every
// thread of that workgroup; the ticket counter is back at zero for the next kernel that uses it.  Atomics on one
// address are served at ~20 ns apiece: kernels that end this way run on a few hundred workgroups at most (grid-stride).
__device__ __forceinline__ bool last_workgroup_out(int *ticket, int n_workgroups)  // n_workgroups: how many draw a ticket
{
    // No agent-scope fence here: on this chip a release fence writes the XCD's whole L2 back, once per workgroup (measured:
    // a 0.76 M-particle slab step 380 -> 510 us).  What the epilogue reads of the other workgroups' work are counters
    // kept by RETURNING atomics (performed at the coherence point before their wave went on) and relaxed agent-scope
    // stores by the ticket-drawing thread itself, which the s_waitcnt below has seen acknowledged.
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int drawn = atomicAdd(ticket, 1);
        s_last = drawn == n_workgroups - 1 ? 1 : 0;
        if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_last != 0;
}
__device__ __forceinline__ int peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one slot of a shared counter per accepted lane, one atomic per wavefront (750 k single atomics on one address were
// 200 us of every re-binning step of a 0.76 M-particle slab)
__device__ __forceinline__ int wave_take_slot(int *counter, bool take)
{
    const unsigned long long m = __ballot(take);
    if (!m) return 0;
    const int lane = threadIdx.x & 63, lead = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == lead) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, lead);
    return base + __popcll(m & (lane ? (~0ull >> (64 - lane)) : 0ull));
}

// Global maxima known (vd_global = all-reduced {max |v|, max drift}): decide whether this step ends with a re-binning,
// write message A, advance the clock.  Every workgroup takes the decision by itself, from values the kernel's own epilogue
// never changes: the run flag of this parity, the all-reduced drift, and Clock::pos_q[q] -- the copy of pos_count that the
// PREVIOUS slot's clock update left for this one.  (On a frozen step only the first n_ticket workgroups draw a ticket, so
// the last of them may advance the clock while a workgroup beyond n_ticket has not even been dispatched yet -- slabs of an
// in-process ring share the chip; read from pos_count itself such a latecomer would see the NEW count, take the re-binning
// branch on the step before a scheduled re-binning and pollute counters, histogram and ticket.)
//   frozen step    : the new state of the send-list particles, in list order;
//   re-binning step: every owned particle's new state goes to `keep` (it stays in this window) and, near a boundary,
//                    to the neighbour -- as in k_slab_pack.
// sn = the NEW state (S[1-q]); layout arrays (mass, id, cell) are those of the step.
__global__ __launch_bounds__(kBlock) void k_slab_pack3(Clock *clk, int q, Grid g, Phys ph, FluidSet sn, SlabPack p, SlabLists L,
                                                       const double *vd_global, int *flags, double half_skin, int K, int *ticket)
{
    const bool run = clk->run[q] != 0;
    if (!run) {  // a stopped loop keeps the following slots idle and sends -1
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            clk->run[1 - q] = 0;
            p.send_l[0] = -1.0;
            p.send_r[0] = -1.0;
        }
        return;
    }
    const double drift = vd_global[1];
    const bool rb = !(drift <= half_skin) || clk->pos_q[q] >= K - 1;  // = clock_step's decision (dyn_K = K)
    const int stride = (int)gridDim.x * kBlock;
    int n_ticket = (int)gridDim.x;
    if (!rb) {
        n_ticket = min(n_ticket, (2 * p.msg_cap + kBlock - 1) / kBlock);
        if ((int)blockIdx.x >= n_ticket) return;  // nothing to do and nothing to wait for
        for (int e = blockIdx.x * kBlock + threadIdx.x; e < 2 * p.msg_cap; e += stride) {
            const int side = e < p.msg_cap ? 0 : 1, sl = e - side * p.msg_cap;
            if (sl >= L.send_cnt[side]) continue;
            const int k = L.send_idx[side][sl];
            const double2 pn = sn.pos[k], vn = sn.vel[k];
            msg_put(side ? p.send_r : p.send_l, p.msg_cap, sl, pn.x + (side ? p.shift_r : p.shift_l), pn.y, vn.x, vn.y,
                    sn.drho[k], 0.0, 0);
        }
    } else {
        // Every workgroup takes one contiguous chunk of the slots.  The kept particles of a chunk get consecutive places
        // in the keep arrays: count them, reserve the range with ONE atomic per workgroup, then hand the places out with a
        // block scan per sweep (a place per wavefront and sweep meant 12 k returning atomics on one address, ~13 ns each:
        // 150 us of every re-binning step of a 0.76 M-particle slab; the few boundary particles keep theirs).
        __shared__ int s_wave[kBlock / 64 + 1];
        __shared__ int s_base;
        const int n = clk->n;
        const int chunk = ((n + (int)gridDim.x - 1) / (int)gridDim.x + kBlock - 1) / kBlock * kBlock;
        const int lo = min(n, (int)blockIdx.x * chunk), hi = min(n, lo + chunk);
        auto kept = [&](int i) {
            if (i >= hi || !owns(g, 0.0, sn.cell[i])) return false;  // (halo copies: their owner sends fresh ones)
            const double x = sn.pos[i].x;
            return x >= p.win_lo && x < p.win_hi;
        };
        int mine_cnt = 0;
        for (int i = lo + (int)threadIdx.x; i < hi; i += kBlock) mine_cnt += kept(i) ? 1 : 0;
        int total;
        (void)block_exclusive_scan_t<kBlock>(mine_cnt, total, s_wave);
        if (threadIdx.x == 0) s_base = total > 0 ? atomicAdd(&p.counters[0], total) : 0;
        __syncthreads();
        int place = s_base;
        for (int base = lo; base < hi; base += kBlock) {  // (uniform trip count per workgroup)
            const int i = base + (int)threadIdx.x;
            const bool mine = i < hi && owns(g, 0.0, sn.cell[i]);
            double2 pn = make_double2(0.0, 0.0), vn = pn;
            double dr = 0.0, m = 0.0;
            int id = 0;
            if (mine) { pn = sn.pos[i]; vn = sn.vel[i]; dr = sn.drho[i]; m = sn.mass[i]; id = sn.id[i]; }
            const double xn = pn.x, yn = pn.y;
            const bool keep = mine && xn >= p.win_lo && xn < p.win_hi;
            const bool to_l = mine && xn < g.own_lo + p.halo_w, to_r = mine && xn >= g.own_hi - p.halo_w;
            int sweep_total;
            const int kk = place + block_exclusive_scan_t<kBlock>(keep ? 1 : 0, sweep_total, s_wave);
            place += sweep_total;
            const int kl = wave_take_slot(&p.counters[1], to_l);
            const int kr = wave_take_slot(&p.counters[2], to_r);
            if (keep) {
                if (kk < p.keep_cap) {
                    p.kpos[kk] = pn; p.kvel[kk] = vn; p.kdrho[kk] = dr; p.kmass[kk] = m; p.kid[kk] = id;
                    int cx, cy;
                    cell_of(g, xn, yn, cx, cy);
                    const int c = cx * g.ncy + cy;
                    p.cellid[kk] = c;
                    atomicAdd(&p.count[c], 1);
                } else atomicOr(flags, 2);
            }
            if (to_l) {
                if (kl < p.msg_cap) msg_put(p.send_l, p.msg_cap, kl, xn + p.shift_l, yn, vn.x, vn.y, dr, m, id);
                else atomicOr(flags, 2);
            }
            if (to_r) {
                if (kr < p.msg_cap) msg_put(p.send_r, p.msg_cap, kr, xn + p.shift_r, yn, vn.x, vn.y, dr, m, id);
                else atomicOr(flags, 2);
            }
        }
    }
    if (!last_workgroup_out(ticket, n_ticket) || threadIdx.x != 0) return;
    p.send_l[0] = (double)(rb ? min(peek(&p.counters[1]), p.msg_cap) : L.send_cnt[0]);
    p.send_r[0] = (double)(rb ? min(peek(&p.counters[2]), p.msg_cap) : L.send_cnt[1]);
    const Clock c0 = *clk;
    clock_step(clk, c0, q, ph, vd_global[0], peek(flags), -1, drift, 0, half_skin, K);
    if (clk->rebuild_now) clk->rebuild_now = 1;  // (no histogram is taken ahead of time in a slab)
}

// Message A arrived, and with it message B: the ids of the send lists the neighbours made in the PREVIOUS step (header
// -1: the lists of the cycle stay).  New ids are resolved to slots first (where do the particles the neighbours will
// keep sending live here?) -- not tied to the step slot being active: the slot after the loop's last step still
// delivers the ids that step made.  Then message A.  Frozen step: scatter the entries into their slots of the new
// state.  Re-binning step: append them behind the kept particles and bin them (k_slab_unpack); the last workgroup out
// then opens the new layout (particle count, counters back to zero).
__global__ __launch_bounds__(kBlock) void k_slab_unpack3(Clock *clk, int q, Grid g, FluidSet sn, SlabPack p, SlabLists L,
                                                         const double *recv_l, const double *recv_r, const int *ids_l,
                                                         const int *ids_r, int *n_new, int *flags, int *ticket)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int cap = p.msg_cap;
    const bool run = clk->run[q] != 0, rb = clk->rebuild_now != 0;
    const int n_now = clk->n;
    const int side = i < cap ? 0 : 1, sl = i - side * cap;
    int slot = -1, n_list = 0;
    if (i < 2 * cap) {
        const int *ids = side ? ids_r : ids_l;
        const int cnt = ids[0];
        if (cnt >= 0) {  // new lists
            n_list = min(cnt, cap);
            if (sl == 0) {
                if (cnt > cap) atomicOr(flags, 2);
                L.recv_cnt[side] = n_list;
            }
            if (sl < n_list) {
                const int id = ids[1 + sl];
                const int k = L.slot_of_id[id];
                if (k < 0 || k >= n_now || sn.id[k] != id) atomicOr(flags, 2);  // I do not hold that particle
                else { L.recv_slot[side][sl] = k; slot = k; }
            }
        } else {
            n_list = L.recv_cnt[side];
            if (sl < n_list) slot = L.recv_slot[side][sl];
        }
    }
    if (run && !rb) {
        if (i < 2 * cap) {
            const double *msg = side ? recv_r : recv_l;
            if (sl == 0 && (int)msg[0] != n_list) atomicOr(flags, 2);  // the neighbour disagrees with my lists
            if (slot >= 0) {
                const double *b = msg + 1;
                sn.pos[slot] = make_double2(b[sl], b[(size_t)cap + sl]);
                sn.vel[slot] = make_double2(b[2 * (size_t)cap + sl], b[3 * (size_t)cap + sl]);
                sn.drho[slot] = b[4 * (size_t)cap + sl];
            }
        }
    } else if (run) {
        const int nl = (int)recv_l[0], nr = (int)recv_r[0];
        const int nk = min(peek(&p.counters[0]), p.keep_cap);
        if (nl < 0 || nr < 0 || nk + nl + nr > p.keep_cap) {
            if (i == 0) { atomicOr(flags, 2); __hip_atomic_store(n_new, min(nk, p.keep_cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        } else {
            if (i == 0) __hip_atomic_store(n_new, nk + nl + nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i < nl + nr) {
                const double *b = (i < nl ? recv_l : recv_r) + 1;
                const int s2 = i < nl ? i : i - nl;
                const int d = nk + i;
                const double x = b[s2], y = b[(size_t)cap + s2];
                p.kpos[d] = make_double2(x, y);
                int cx, cy;
                cell_of(g, x, y, cx, cy);
                const int c = cx * g.ncy + cy;
                p.cellid[d] = c;
                atomicAdd(&p.count[c], 1);
                p.kvel[d] = make_double2(b[2 * (size_t)cap + s2], b[3 * (size_t)cap + s2]);
                p.kdrho[d] = b[4 * (size_t)cap + s2];
                p.kmass[d] = b[5 * (size_t)cap + s2];
                p.kid[d] = (int)b[6 * (size_t)cap + s2];
            }
        }
    }
    if (!(run && rb)) return;  // (uniform over the grid: nobody takes a ticket)
    if (!last_workgroup_out(ticket, (int)gridDim.x) || threadIdx.x != 0) return;
    clk->n = peek(n_new);  // from here on the re-binning chain works on the new particle count
    p.counters[0] = 0; p.counters[1] = 0; p.counters[2] = 0;
    L.send_cnt[0] = 0; L.send_cnt[1] = 0;
}

// The local maxima a skinned slab feeds into the step's all-reduce -- max |v| of the kicked velocities and max drift of the
// owned particles -- as a kernel of its own on the slab's SECOND stream: both exist once pass CD is through, so the reduction
// and the all-reduce behind it run beside pass E instead of behind it (round 3: a tail workgroup of pass E, the all-reduce
// exposed on the step's only stream).  out[0..1] = {max |v|, max drift}; a stopped loop reports zeros.
constexpr int kSlabMaxBlocks = 256;
__global__ __launch_bounds__(kBlock) void k_slab_maxima(const Clock *clk, int q, Grid g, int n_vpart, const double *dpart,
                                                        const double2 *veln, const int *cell, double *part, double *out,
                                                        int *ticket)
{
    const bool run = clk->run[q] != 0;
    double m = 0.0, d = 0.0;
    if (run) {
        const int n = clk->n, stride = (int)gridDim.x * kBlock;
        // (four particles per trip, their loads requested together and unconditionally: the loop is a chain of memory round
        //  trips -- 23 of them per thread with one particle per trip cost 38 us at 0.76 M particles)
        for (int i0 = blockIdx.x * kBlock + threadIdx.x; i0 < n; i0 += 4 * stride) {
            int ce[4];
            double2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * stride, n - 1);
                ce[u] = cell[i];
                v[u] = veln[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * stride >= n || !owns(g, 0.0, ce[u])) continue;  // (skinned slabs own by the binned column)
                double v2 = v[u].x * v[u].x + v[u].y * v[u].y;
                if (v2 != v2) v2 = INFINITY;  // NaN poisons the maximum on purpose
                m = fmax(m, v2);
            }
        }
        for (int k = blockIdx.x * kBlock + threadIdx.x; k < n_vpart; k += gridDim.x * kBlock) d = fmax(d, dpart[k]);
    }
    m = wave_max(m);
    d = wave_max(d);
    __shared__ double s_m[kBlock / 64], s_d[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_d[threadIdx.x >> 6] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / 64; ++k) { m = fmax(m, s_m[k]); d = fmax(d, s_d[k]); }
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&part[2 * blockIdx.x]), (unsigned long long)__double_as_longlong(m),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&part[2 * blockIdx.x + 1]), (unsigned long long)__double_as_longlong(d),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the last workgroup out folds the partial maxima -- one per thread: a single thread reading them with agent-scope loads
    // one after the other took 80 us (the compiler does not overlap atomic loads)
    if (!last_workgroup_out(ticket, (int)gridDim.x)) return;
    m = 0.0; d = 0.0;
    for (int b = (int)threadIdx.x; b < (int)gridDim.x; b += kBlock) {
        m = fmax(m, __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(&part[2 * b]),
                                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        d = fmax(d, __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(&part[2 * b + 1]),
                                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    }
    m = wave_max(m);
    d = wave_max(d);
    __syncthreads();  // (s_m / s_d are reused)
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_d[threadIdx.x >> 6] = d; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int k = 1; k < kBlock / 64; ++k) { m = fmax(m, s_m[k]); d = fmax(d, s_d[k]); }
    out[0] = run ? sqrt(m) : 0.0;
    out[1] = run ? sqrt(d) : 0.0;
}

__global__ __launch_bounds__(kBlock) void k_slot_of_id(int n, const int *id, int *slot_of_id)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) slot_of_id[id[i]] = i;
}

// re-binning steps, on the NEW layout: owned particles within halo_w of a boundary form the send lists of the cycle;
// their ids are message B, whose header the last workgroup out writes (-1: the lists of the cycle stay)
__global__ __launch_bounds__(kBlock) void k_slab_sendlist(const Clock *clk, int q, Grid g, FluidSet sn, SlabPack p, SlabLists L,
                                                          int *flags, int force, int *ticket)
{
    const bool rb = force || (clk->run[q] && clk->rebuild_now);  // force: the lists of the first cycle
    if (!rb) {  // (uniform over the grid)
        if (blockIdx.x == 0 && threadIdx.x == 0) { L.ids_send[0][0] = -1; L.ids_send[1][0] = -1; }
        return;
    }
    const int n = clk->n;
    for (int base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {  // (uniform trip count per workgroup)
        const int i = base + (int)threadIdx.x;
        const bool mine = i < n && owns(g, 0.0, sn.cell[i]);
        const double x = mine ? sn.pos[i].x : 0.0;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const bool near = mine && (side ? x >= g.own_hi - p.halo_w : x < g.own_lo + p.halo_w);
            const int k = wave_take_slot(&L.send_cnt[side], near);
            if (!near) continue;
            if (k < p.msg_cap) {
                L.send_idx[side][k] = i;
                L.ids_send[side][1 + k] = sn.id[i];
            } else atomicOr(flags, 2);
        }
    }
    if (!last_workgroup_out(ticket, (int)gridDim.x) || threadIdx.x != 0) return;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int cnt = min(peek(&L.send_cnt[side]), p.msg_cap);
        L.send_cnt[side] = cnt;
        L.ids_send[side][0] = cnt;
    }
}

// message B, receiving side: where do the particles the neighbours will keep sending live here?  The ids travel with the
// NEXT step's message A (one exchange per step instead of two) and are looked at before that message is unpacked; a header
// of -1 says the lists of the cycle stay.  Not tied to the step slot being active: the slot after the loop's last step
// still delivers the ids that step made.
__global__ __launch_bounds__(kBlock) void k_slab_recvslots(const Clock *clk, int q, SlabPack p, SlabLists L, const int *ids_l,
                                                           const int *ids_r, const int *id_of_slot, int *flags, int force)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= 2 * p.msg_cap) return;
    const int side = i < p.msg_cap ? 0 : 1, sl = i - side * p.msg_cap;
    const int *ids = side ? ids_r : ids_l;
    const int cnt = ids[0];
    if (!force && cnt < 0) return;
    if (sl == 0) {
        if (cnt < 0 || cnt > p.msg_cap) atomicOr(flags, 2);  // (force: the first lists of a run must be there)
        L.recv_cnt[side] = max(0, min(cnt, p.msg_cap));
    }
    if (sl >= cnt || sl >= p.msg_cap) return;
    const int id = ids[1 + sl];
    const int k = L.slot_of_id[id];
    if (k < 0 || k >= clk->n || id_of_slot[k] != id) { atomicOr(flags, 2); return; }  // I do not hold that particle
    L.recv_slot[side][sl] = k;
}

}  // namespace sphx
