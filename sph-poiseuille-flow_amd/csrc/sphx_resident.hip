// sphx_resident.hip -- host side of the device-resident SPH step (include/sphx.h section 2):
// context creation (cell grid build on device), the step loop as hipGraph replays, download, monitors,
// MEX-convention pair-list emission, per-kernel HIP-event timing, and the x-slab (multi-GPU) entry
// points.  The kernels are in sphx_kernels.hpp.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is loaded at run time (see Rccl), libsphx does not link it

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>

#include "sphx_common.hpp"
#include "sphx_kernels.hpp"

namespace sphx {

struct KernelTimer {
    std::vector<std::string> names;
    std::vector<double> total_ms;
    std::vector<int64_t> launches;
    struct Pending { int idx; hipEvent_t a, b; };
    std::vector<Pending> pending;     // eager launches: one-shot events
    std::vector<Pending> graph_evs;   // events recorded by nodes of the profiling graph (re-armed per replay)
    bool warned = false, timer_log = false;  // (timer_log: SPHX_DEBUG_SWITCHES=log)
    int index_of(const char *name)
    {
        for (size_t k = 0; k < names.size(); ++k) if (names[k] == name) return (int)k;
        names.emplace_back(name); total_ms.push_back(0.0); launches.push_back(0);
        return (int)names.size() - 1;
    }
    void add(int idx, hipEvent_t a, hipEvent_t b)
    {
        float ms = 0.f;
        const hipError_t e = hipEventElapsedTime(&ms, a, b);
        if (e == hipSuccess) { total_ms[idx] += ms; launches[idx] += 1; }
        else {
            (void)hipGetLastError();
            if (timer_log && !warned) { warned = true; fprintf(stderr, "sphx: hipEventElapsedTime: %s\n", hipGetErrorString(e)); }
        }
    }
    void collect()
    {
        for (auto &p : pending) { add(p.idx, p.a, p.b); (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        pending.clear();
    }
    void collect_graph() { for (auto &p : graph_evs) add(p.idx, p.a, p.b); }
    void drop_graph_events()
    {
        for (auto &p : graph_evs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        graph_evs.clear();
    }
};

}  // namespace sphx

using namespace sphx;

struct sphx_ctx {
    sphx_params prm{};
    Grid grid{};
    Phys phys{};
    int nf = 0, nw = 0, nt = 0;  // caller's global counts (single GPU: nf particles resident)
    int cap = 0;                 // particle capacity of the device arrays
    int lpp = 1, spg = 2;
    bool big_scan = false;
    int n_vpart = 0;             // entries of vpart / dpart the clock kernel reduces
    int n_tiles = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;

    // Cell grid with a skin: cells are 2h + skin wide, particles are re-binned only every `rebuild_every` steps
    // ("rebuild" step: 7 launches, "move" step: 5).  Between two builds every sweep is centred on the cell a
    // particle was binned into, which finds all neighbours while nobody has drifted more than skin/2; the
    // device clock tracks the largest drift (Clock::drift) and stops the loop before the bound is violated.
    double skin = 0.0;
    int rebuild_every = 1;       // re-binning interval K (constant)
    int64_t cool_until = 0;      // after a forced rebuild every step up to this step index re-bins (cool-down)
    int64_t cool_len = 0;        // length of the last cool-down (doubles when forced rebuilds keep coming)
    int64_t prov_step = 0;       // step index the next enqueued slot will have if every slot before it executes

    // where the current state lives: state buffers S[cur] (x,y,vx,vy,drho), layout buffers L[lay]
    // (mass,id,start,cell); `pos` = steps taken since the grid was built.  Derived from the device step count.
    int cur = 0, lay = 0, pos = 0;
    int64_t epoch_step = 0;      // step count at which (epoch_cur, epoch_lay, epoch_pos, ...) held
    int epoch_cur = 0, epoch_lay = 0, epoch_pos = 0;
    int epoch_out_lay = 0;
    int out_lay = 0;             // layout the per-step outputs (rho,p,force,Vol,B) are stored in; when it is not
                                 // `lay`, tmp.src_of maps current slots to the slots of those outputs
    int64_t n_forced_rebuilds = 0, last_forced_step = 0;
    int64_t n_rebins = 0, epoch_n_rebins = 0;  // re-binnings executed by step slots of the static schedule (not the forced ones)
    int64_t pending_target = 0;  // step count the sphx_ctx_enqueue_steps calls since the last sync aim for
    bool have_step_outputs = false;

    // storage
    DevBuf<double2> fpos_[2], fvel_[2], fposb_[2];
    DevBuf<double> fdrho_[2], fmass_[2];
    DevBuf<int> fid_[2], fstart_[2], fcell_[2];
    DevBuf<double2> posn, veln, ffp, ff;
    DevBuf<double4> fa, fB;
    DevBuf<double> drhon, rho_out, p_out, vpart, dpart, vtile, fvol;
    int n_vtiles = 0;            // > 0: k_max_tiles folds the per-block maxima first (very many blocks)
    DevBuf<int> cellid, count, perm, src_of, nl_idx, nl_cnt, sl_idx, sl_cnt, flags, tile;
    DevBuf<int> nl_pk, sl_pk, nl_pk2;  // large-channel kernels: fluid entries as 16-bit index differences (FluidTmp::nl_pk)
    DevBuf<int> tmap;                  // ... and every workgroup's tile layout (FluidTmp::tmap)
    DevBuf<double2> wpos;
    DevBuf<double4> wa;
    DevBuf<int> wid, wstart, wrow_any;
    DevBuf<Clock> clock;
    DevBuf<double> tau_part, tau_out;
    Clock *h_clock = nullptr;  // pinned
    Clock *h_pub = nullptr;    // pinned + mapped: the device writes the clock here whenever the loop stops (Clock::pub)
    long long host_seq = 0;    // batches armed so far (k_prepare launches), cf. Clock::seq

    FluidTmp tmp{};
    Walls walls{};
    int n_blocks_particles = 0;  // grid of the LPP kernels (capacity based)
    int n_blocks_flat = 0;       // grid of one-thread-per-particle kernels

    // Replayable graphs, keyed by the phase they were captured from and their length: {cur, lay, pos, slots}.  A graph
    // of graph_slots() slots (a multiple of the 2K-step period) hands the phase back unchanged, so a long run replays
    // ONE graph whatever phase it was entered at; exact-length batches (sphx_ctx_enqueue_steps, advance with
    // max_steps -- a caller logging every 20 steps) get a graph per (phase, length), captured the first time that
    // combination comes up (or ahead of time by sphx_ctx_prepare_steps).
    struct CachedGraph { hipGraphExec_t exec = nullptr; bool launched = false; };
    std::map<std::array<int, 4>, CachedGraph> graphs;
    int64_t slots_replayed = 0, slots_eager = 0, graphs_captured = 0;
    int64_t chunk_slots = 128;   // slots enqueued between two host looks at the clock (adaptive, see advance)
    bool profiling = false;
    KernelTimer timer;

    // pair list held for sphx_neighbor_fetch
    DevBuf<double> pl_i, pl_j, pl_dx, pl_dy, pl_r, pl_W, pl_dW;
    size_t pl_n = 0;
    bool pl_valid = false;

    // x-slab state
    bool is_slab = false;
    int rank = 0, n_ranks = 1, halo_cols = 0, msg_cap = 0;
    int col0 = 0, col1 = 0;  // owned global columns [col0, col1)
    DevBuf<double2> kpos, kvel;
    DevBuf<double> kdrho, kmass;
    DevBuf<int> kid, counters, n_new, ticket;  // (ticket: see last_workgroup_out)
    SlabPack pack{};
    int64_t slab_steps_enqueued = 0, slab_step0 = 0;
    // native step loop (sphx_slab_run / sphx_slab_group_run): library-owned message buffers, RCCL communicator
    DevBuf<double> msg_sl, msg_sr, msg_rl, msg_rr, vmax_l, vmax_g;  // (vmax_*: {max |v|, max drift})
    // skinned slabs (re-binning every K-th step): the exchange lists of the cycle, see SlabLists
    DevBuf<int> send_idx_[2], recv_slot_[2], send_cnt, recv_cnt, slot_of_id, ids_s_[2], ids_r_[2];
    SlabLists lists{};
    bool lists_ready = false;
    ncclComm_t comm = nullptr;
    hipEvent_t ev_computed = nullptr, ev_received = nullptr;  // single-process ring: cross-stream ordering
    // Skinned slabs, native loops: the local maxima and the all-reduce behind them run on a second stream beside pass E
    // (k_slab_maxima): fork after pass CD (ev_cd), join in front of k_slab_pack3 (ev_ar); ev_max: "my local maxima are out"
    // (in-process rings, where a kernel stands in for the all-reduce)
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_cd = nullptr, ev_ar = nullptr, ev_max = nullptr;
    DevBuf<double> max_part;
    DevBuf<int> ticket2;
    hipEvent_t ev_p = nullptr, ev_u = nullptr;  // "message A is packed" / "the halo is refreshed (or the layout rebuilt)"
    int pass_a_part = 0;           // what the next launch of pass A's walk covers: 0 all, 1 interior, 2 boundary workgroups (slab_part_skips)
    bool a_interior_done = false;  // the interior workgroups of the coming step's pass A were launched behind this step's pack3
    // Whole slab steps as ONE replayable graph (sphx_slab_graph_prepare): kSlabGraphSteps steps of the native loop -- kernels,
    // the RCCL calls (sphx_slab_run) or the device-to-device copies and cross-stream dependencies of an in-process ring
    // (sphx_slab_group_run; held by slab 0) -- captured once the loop has run eagerly at least twice
    hipGraphExec_t steps_graph = nullptr;
    std::vector<const sphx_ctx *> steps_graph_ring;  // the contexts the graph was captured for
    int steps_graph_cur = 0;                          // ... and the state parity it starts from
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    // Dynamic re-binning (large channels): the device decides when to re-bin, every step carries the (self-skipping)
    // re-binning kernels, the layout is rebuilt in place -> lay stays 0, only the state parity alternates
    bool dyn = false;
    // Small channels: pass E of a step that does not re-bin and pass A of the next step share one launch
    // (k_continuity_density).  The neighbour list and the {Vol, p, rho_h, rho} records then exist once per state parity:
    // tmp_par[p] is `tmp` with the buffers of parity p; out_par = the parity of the last executed step (its Vol).
    bool fuse_ea = false;
    bool lds_tiles_a = false;    // pass A's walk gathers the candidate positions from an LDS tile
    bool sweep_kernels = false;  // pass A's cell sweep in its large-channel form (k_density_sweep_w)
    int n_in = 1;                // dual-rate loop: inner sub-steps per step slot (1 = the reference's single-rate loop)
    DevBuf<double2> vel2;        // ... and the second velocity array its sub-steps alternate with
    DevBuf<double4> fa2;
    DevBuf<double> fvol2;
    DevBuf<int> nl_idx2, nl_cnt2;
    FluidTmp tmp_par[2] = {};
    int out_par = 0, epoch_out_par = 0;
    bool walk_kernels = false;   // lanes_per_particle <= 8: passes B, CD, E run their large-channel ("_w") forms
    bool lds_tiles = false;      // ... and the force pass stages its tile's neighbourhood in LDS
    bool lds_tiles_be = false;   // ... KGC and continuity too (2 lanes per particle, channel larger than the Infinity Cache)
    bool coded_lists = false;    // ... and the lists name tile slots instead of index differences (kSlotCodes, sphx_kernels.hpp)
    bool tail_clock = false;     // move steps carry their clock update in a tail workgroup of pass E (small channels)

    FluidSet view(int q, int l)
    {
        return FluidSet{fpos_[q].get(), fvel_[q].get(), fdrho_[q].get(), fmass_[l].get(), fid_[l].get(),
                        fstart_[l].get(), fcell_[l].get(), skin > 0.0 ? fposb_[l].get() : nullptr};
    }
    double half_skin() const { return 0.5 * skin; }
    unsigned long long *vpart_reset() const
    {
        return tail_clock ? reinterpret_cast<unsigned long long *>(vpart.get()) : nullptr;
    }

    void drop_graph()
    {
        for (auto &kv : graphs)
            if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        graphs.clear();
    }

    ~sphx_ctx()
    {
        if (stream) (void)hipStreamSynchronize(stream);  // the buffers go back to the pool: nothing may still use them
        drop_graph();
        if (steps_graph) (void)hipGraphExecDestroy(steps_graph);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        timer.collect();
        timer.drop_graph_events();
        if (ev_computed) (void)hipEventDestroy(ev_computed);
        if (ev_received) (void)hipEventDestroy(ev_received);
        if (stream2) (void)hipStreamSynchronize(stream2);
        if (ev_cd) (void)hipEventDestroy(ev_cd);
        if (ev_ar) (void)hipEventDestroy(ev_ar);
        if (ev_max) (void)hipEventDestroy(ev_max);
        if (ev_p) (void)hipEventDestroy(ev_p);
        if (ev_u) (void)hipEventDestroy(ev_u);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (h_clock) (void)hipHostFree(h_clock);
        if (h_pub) (void)hipHostFree(h_pub);
        if (stream && own_stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

// A/B switches for measurements, all behind ONE environment variable read once per process:
//   SPHX_DEBUG_SWITCHES=no_tail_clock,no_fuse_ea,no_lds_tiles,no_coded_lists,no_lazy_out,forces_tile_320,log
// (no_tail_clock: the clock update as a launch of its own on every step; no_fuse_ea: passes E and A in separate launches;
//  no_lds_tiles: large-channel passes gather from global memory; no_coded_lists: index differences in every list, never tile
//  slots; no_lazy_out: force, force_prior, rho, p written by every step (FluidTmp::lazy_out); log: forced re-binnings and timer
//  problems on stderr)
struct DebugSwitches {
    bool no_tail_clock = false, no_fuse_ea = false, no_lds_tiles = false, no_coded_lists = false, no_lazy_out = false, log = false;
    bool no_slab_overlap = false;  // skinned slabs: the local maxima from pass E's tail workgroup, the all-reduce on the step's only stream (round 3)
    bool full_copyback = false;  // dynamic contexts: the in-place re-binning copies the whole layout back (round 3)
    int tail_limit = 0;  // > 0: largest pass (in workgroups) whose clock update rides in pass E's tail workgroup
    int forces_tile = 0;    // 320: the old tile size of the slot-coded force pass
    int tiles_be_from = 0;  // > 0: passes B, E and A stage LDS tiles from this many resident particles (2 lanes per particle)
};
const DebugSwitches &debug_switches()
{
    static const DebugSwitches sw = [] {
        DebugSwitches d;
        const char *e = std::getenv("SPHX_DEBUG_SWITCHES");
        const std::string v = e ? e : "";
        auto has = [&](const char *name) { return ("," + v + ",").find(std::string(",") + name + ",") != std::string::npos; };
        d.no_tail_clock = has("no_tail_clock");
        d.no_fuse_ea = has("no_fuse_ea");
        d.no_lds_tiles = has("no_lds_tiles");
        d.no_coded_lists = has("no_coded_lists");
        d.no_lazy_out = has("no_lazy_out");
        d.log = has("log");
        d.no_slab_overlap = has("no_slab_overlap");
        d.full_copyback = has("full_copyback");
        for (int lim : {1024, 2048, 4096, 8192, 16384})
            if (has(("tail_limit_" + std::to_string(lim)).c_str())) d.tail_limit = lim;
        for (int from : {1, 250000, 500000, 750000, 1000000, 1500000, 3000000})
            if (has(("tiles_be_from_" + std::to_string(from)).c_str())) d.tiles_be_from = from;
        if (has("forces_tile_320")) d.forces_tile = 320;
        return d;
    }();
    return sw;
}

// largest pass, in workgroups, whose clock update rides in a tail workgroup of pass E (and whose passes E and A share a launch)
// Round 3: 2 048 -> 4 096.  Fusing E and A and dropping the clock launch still pays at 0.3-0.5 M particles (2 lanes: 324 k
// 139.7 -> 129.3 us/step, C4 = 499 k 180.5 -> 172.2); at 0.83 M (6 500 workgroups) it is neutral to slightly worse (270 -> 271.5).
constexpr int kTailClockBlocks = 4096;
int tail_clock_limit() { return debug_switches().tail_limit > 0 ? debug_switches().tail_limit : kTailClockBlocks; }

thread_local sphx_ctx *g_search_ctx = nullptr;  // owns the temporary context of sphx_neighbor_search until fetch
thread_local sphx_ctx *g_fetch_src = nullptr;   // context whose pair list the next sphx_neighbor_fetch copies out

template <typename K, typename... Args>
void launch_s(sphx_ctx *c, const char *name, K kernel, dim3 grid, dim3 block, size_t shmem, Args... args)
{
    if (c->profiling) {  // eager launch with an event pair (event-record nodes in a replayed graph read 0 on ROCm 7.2)
        KernelTimer::Pending p;
        p.idx = c->timer.index_of(name);
        SPHX_HIP(hipEventCreate(&p.a));
        SPHX_HIP(hipEventCreate(&p.b));
        SPHX_HIP(hipEventRecord(p.a, c->stream));
        hipLaunchKernelGGL(kernel, grid, block, shmem, c->stream, args...);
        SPHX_HIP(hipEventRecord(p.b, c->stream));
        c->timer.pending.push_back(p);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shmem, c->stream, args...);
    }
}

template <typename K, typename... Args>
void launch(sphx_ctx *c, const char *name, K kernel, dim3 grid, dim3 block, Args... args)
{
    launch_s(c, name, kernel, grid, block, 0, args...);
}

// gather of the persistent fields (pos, vel, drho, mass) into destination view d
ReorderArgs reorder_args(const double2 *pos, const double2 *vel, const double *drho, const double *mass, const int *id_src,
                         const FluidSet &d, int *src_of, int *slot_of_id = nullptr)
{
    ReorderArgs ra{};
    ra.n2 = 2;
    ra.src2[0] = pos; ra.dst2[0] = d.pos;
    ra.src2[1] = vel; ra.dst2[1] = d.vel;
    if (d.posb) {  // remember where every particle was when it was binned
        ra.n2 = 3;
        ra.src2[2] = pos; ra.dst2[2] = d.posb;
    }
    ra.n1 = 2;
    ra.src1[0] = drho; ra.dst1[0] = d.drho;
    ra.src1[1] = mass; ra.dst1[1] = d.mass;
    ra.id_src = id_src;
    ra.id_dst = d.id;
    ra.src_of = src_of;
    ra.cell_dst = d.cell;
    ra.slot_of_id = slot_of_id;
    return ra;
}

// Inner sub-steps per step slot a context asking for the dual-rate loop would run (1 = single rate).
int dual_rate_substeps(const sphx_params &prm)
{
    if (prm.dual_rate <= 1) return 1;
    const double h = prm.h;
    const double dt_ac = 0.25 * h / (1.1 * prm.c_f);  // at max|v| = 0.1 c_f, where the reference's set-up runs
    const double dt_visc = 0.125 * h * h * prm.rho0 / std::max(prm.mu, 1e-12);
    const double dt_body = 0.25 * std::sqrt(h / std::max(std::fabs(prm.gravity_g), 1e-12));
    const int fit = (int)std::floor(std::min(dt_visc, dt_body) / dt_ac);
    return std::max(1, std::min(fit, (int)prm.dual_rate));
}

// The four neighbour passes on state view `s`, writing the end-of-step state through t.posn / veln / drhon.
// only: 0 = all four, 1..4 = just density / kgc / forces / continuity (kernel timing)
// dmode: 0 = pass A sweeps the cells; 1 = sweeps and writes the superset list (first step after a re-bin);
//        2 = walks the superset list
// tail: pass E gets one workgroup more, which advances the clock (no k_clock_scan after this step)
// inner (dual-rate loop, compact kernels only): this launch of pass CD / E belongs to an inner sub-step -- CD does the
//        pressure part only, E hands the next sub-step its half-step density
template <int LPP>
void launch_physics(sphx_ctx *c, int q, const FluidSet &s, const FluidTmp &t, int do_hist, int only, int dmode, int tail, int inner)
{
    const dim3 gp(c->n_blocks_particles), bp(kBlock);
    const Clock *clk = c->clock.get();
    const dim3 ge(c->n_blocks_particles + (tail ? 1 : 0));  // tail: 1 = clock, 2 = a slab's local maxima (slab_seal_tail)
    const char *name_e = tail ? "k_continuity_clock" : "k_continuity";
    if constexpr (LPP >= 16) {
        // small channels: the compact kernels (32-bit lists)
        if (!only || only == 1) {
            if (dmode == 0) launch(c, "k_density", k_density<LPP, 0>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, -1);
            else if (dmode == 1) launch(c, "k_density_build", k_density<LPP, 1>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, -1);
            else {
                // dmode 2: walk the superset list; 3 (dynamic contexts): build and walk, each skipping itself according to the
                // clock's `fresh`
                // (dmode 4, slabs: the walk alone, on a grid that is not fresh; pass_a_part: interior / boundary workgroups only)
                const int cond = dmode == 2 ? -1 : (c->pass_a_part << kPassPartShift);
                if (dmode == 3) launch(c, "k_density_build", k_density<LPP, 1>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, 1);
                launch(c, "k_density_walk", k_density<LPP, 2>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, cond);
            }
        }
        if (!only || only == 2) launch(c, "k_kgc", k_kgc<LPP>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, c->fuse_ea ? 1 : 0);
        if (!only || only == 3) launch(c, inner ? "k_forces_inner" : "k_forces", k_forces<LPP>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, inner);
        if (!only || only == 4)
            launch(c, inner ? "k_continuity_inner" : name_e, k_continuity<LPP, false, 0>, ge, bp, c->clock.get(), q, c->grid, c->phys, s, t, c->walls, do_hist, tail, inner);
    } else {
        // large channels (few lanes per particle): the "_w" forms, see sphx_kernels.hpp -- fluid list entries are 16-bit index
        // differences, so builders and walkers always go together
        constexpr int T = tile_slots(LPP);
        if (!only || only == 1) {
            auto sweep = [&](const char *name, auto mode, int cond) {  // the cell sweep: mode 0 writes the step's list, mode 1 the superset list as well
                constexpr int M = decltype(mode)::value;
                // the build variant of a dynamic context is idle on four steps out of five: a grid-stride launch of an eighth
                // of the workgroups costs an eighth to skip (6 M particles: 47 k idle workgroups were ~70 us of every step)
                const unsigned nb = cond >= 0 ? std::max<unsigned>(1u, (unsigned)c->n_blocks_particles / 8u) : (unsigned)c->n_blocks_particles;
                if constexpr (LPP == 2) {
                    if (c->coded_lists) {
                        launch(c, name, k_density_sweep_w<LPP, M, true>, dim3(nb), bp, clk, q, c->grid, c->phys, s, t, c->walls, cond, c->n_blocks_particles);
                        return;
                    }
                }
                launch(c, name, k_density_sweep_w<LPP, M>, dim3(nb), bp, clk, q, c->grid, c->phys, s, t, c->walls, cond, c->n_blocks_particles);
            };
            if (dmode == 0) sweep("k_density", std::integral_constant<int, 0>{}, -1);
            else if (dmode == 1) sweep("k_density_build", std::integral_constant<int, 1>{}, -1);
            else {
                const int cond = dmode == 2 ? -1 : (c->pass_a_part << kPassPartShift);  // (dmode 4: see the compact kernels)
                if (dmode == 3) sweep("k_density_build", std::integral_constant<int, 1>{}, 1);
                bool done = false;
                if constexpr (LPP == 2) {
                    if (c->coded_lists) {
                        launch(c, "k_density_walk", k_density_w<LPP, kSlotCodes, true>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, cond);
                        done = true;
                    }
                }
                if (done) {}
                else if (c->lds_tiles_a) launch(c, "k_density_walk", k_density_w<LPP, T>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, cond);
                else launch(c, "k_density_walk", k_density_w<LPP, 0>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, cond);
            }
        }
        // LDS tiles (tile_ranges): the force pass always; KGC and continuity where measured to pay (lds_tiles_be)
        bool coded = false;  // slot-coded lists (2 lanes per particle, every pass with a tile): the CODED forms of the same kernels
        if constexpr (LPP == 2) coded = c->coded_lists;
        if (!only || only == 2) {
            if constexpr (LPP == 2) {
                if (coded) launch(c, "k_kgc", k_kgc_w<LPP, kSlotCodes, true>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, c->fuse_ea ? 1 : 0);
            }
            if (coded) {}
            else if (c->lds_tiles_be) launch(c, "k_kgc", k_kgc_w<LPP, T>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, c->fuse_ea ? 1 : 0);
            else launch(c, "k_kgc", k_kgc_w<LPP, 0>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls, c->fuse_ea ? 1 : 0);
        }
        if (!only || only == 3) {
            if constexpr (LPP == 2) {
                // (the whole layout, four workgroups per CU: with 320 slots, five per CU, a quarter of the neighbours came from
                //  global memory in nearly every trip of every wave -- 6 M particles 516 -> 489 us, forces_tile_320 for the old size)
                if (coded) {
                    if (debug_switches().forces_tile == 320) launch(c, "k_forces", k_forces_w<LPP, 320, true>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls);
                    else launch(c, "k_forces", k_forces_w<LPP, kForceSlots, true>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls);
                }
            }
            if (coded) {}
            else if (c->lds_tiles) launch(c, "k_forces", k_forces_w<LPP, (LPP <= 2 ? 320 : T)>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls);
            else launch(c, "k_forces", k_forces_w<LPP, 0>, gp, bp, clk, q, c->grid, c->phys, s, t, c->walls);
        }
        if (!only || only == 4) {
            if constexpr (LPP == 2) {
                if (coded) launch(c, name_e, k_continuity<LPP, true, kSlotCodes, true>, ge, bp, c->clock.get(), q, c->grid, c->phys, s, t, c->walls, do_hist, tail, 0);
            }
            if (coded) {}
            else if (c->lds_tiles_be) launch(c, name_e, k_continuity<LPP, true, T>, ge, bp, c->clock.get(), q, c->grid, c->phys, s, t, c->walls, do_hist, tail, 0);
            else launch(c, name_e, k_continuity<LPP, true, 0>, ge, bp, c->clock.get(), q, c->grid, c->phys, s, t, c->walls, do_hist, tail, 0);
        }
    }
}

void launch_physics_any(sphx_ctx *c, int q, const FluidSet &s, const FluidTmp &t, int do_hist, int only = 0, int dmode = 0,
                        int tail = 0, int inner = 0)
{
    switch (c->lpp) {
        case 1: launch_physics<1>(c, q, s, t, do_hist, only, dmode, tail, inner); break;
        case 2: launch_physics<2>(c, q, s, t, do_hist, only, dmode, tail, inner); break;
        case 4: launch_physics<4>(c, q, s, t, do_hist, only, dmode, tail, inner); break;
        case 8: launch_physics<8>(c, q, s, t, do_hist, only, dmode, tail, inner); break;
        case 16: launch_physics<16>(c, q, s, t, do_hist, only, dmode, tail, inner); break;
        case 32: launch_physics<32>(c, q, s, t, do_hist, only, dmode, tail, inner); break;
        default: throw Error(SPHX_ERR_ARG, "SPHX:Ctx:lpp", "lanes_per_particle must be 1,2,4,8,16 or 32");
    }
}

// exclusive scan of the cell histogram into start_next (three kernels on big grids)
void launch_cell_scan(sphx_ctx *c, const Clock *clk, int q, int *start_next)
{
    if (!c->big_scan) {
        launch(c, "k_scan", k_scan_only, dim3(1), dim3(kScanBlock), clk, q, (const int *)c->count.get(), start_next,
               c->grid.ncells);
    } else {
        int *tile_sum = c->tile.get(), *tile_off = c->tile.get() + c->n_tiles + 1;
        launch(c, "k_scan_tiles", k_scan_tiles, dim3(c->n_tiles), dim3(kScanBlock), clk, q, (const int *)c->count.get(),
               start_next, tile_sum, c->grid.ncells);
        launch(c, "k_scan_sums", k_scan_only, dim3(1), dim3(kScanBlock), clk, q, (const int *)tile_sum, tile_off, c->n_tiles);
        launch(c, "k_scan_add", k_scan_add, dim3(c->n_tiles), dim3(kScanBlock), clk, q, start_next, (const int *)tile_off,
               c->grid.ncells, c->n_tiles);
    }
}

// index -> cell slot, then the gather of the persistent fields into destination view d
void launch_scatter_reorder(sphx_ctx *c, const Clock *clk, int q, const ReorderArgs &ra, const FluidSet &d, int max_blocks = 0)
{
    const dim3 g1(max_blocks > 0 ? std::min(c->n_blocks_flat, max_blocks) : c->n_blocks_flat), bp(kBlock);
    launch(c, "k_scatter", k_scatter, g1, bp, clk, q, 0, (const int *)c->cellid.get(), c->count.get(), (const int *)d.start,
           c->perm.get());
    launch(c, "k_reorder", k_reorder, g1, bp, clk, q, 0, (const int *)c->cellid.get(), (const int *)d.start,
           (const int *)c->perm.get(), ra);
}

template <int LPP>
void launch_fused_ea_t(sphx_ctx *c, int q, const FluidSet &s, const FluidTmp &t, const FluidSet &sn, const FluidTmp &tn, int tail)
{
    launch(c, "k_continuity_density", k_continuity_density<LPP>, dim3(2 * c->n_blocks_particles + tail), dim3(kBlock),
           c->clock.get(), q, c->grid, c->phys, s, t, c->walls, sn, tn, tail);
}
template <int LPP>
void launch_fused_ea_w(sphx_ctx *c, int q, const FluidSet &s, const FluidTmp &t, const FluidSet &sn, const FluidTmp &tn, int tail)
{
    launch(c, "k_continuity_density", k_continuity_density_w<LPP>, dim3(2 * c->n_blocks_particles + tail), dim3(kBlock),
           c->clock.get(), q, c->grid, c->phys, s, t, c->walls, sn, tn, tail);
}
// tail = 0: without the clock workgroup (kernel timing)
void launch_fused_ea(sphx_ctx *c, int q, const FluidSet &s, const FluidTmp &t, const FluidSet &sn, const FluidTmp &tn, int tail = 1)
{
    switch (c->lpp) {  // 16 / 32 lanes per particle: the compact kernels; fewer: their large-channel forms
        case 2: launch_fused_ea_w<2>(c, q, s, t, sn, tn, tail); break;
        case 4: launch_fused_ea_w<4>(c, q, s, t, sn, tn, tail); break;
        case 8: launch_fused_ea_w<8>(c, q, s, t, sn, tn, tail); break;
        case 16: launch_fused_ea_t<16>(c, q, s, t, sn, tn, tail); break;
        case 32: launch_fused_ea_t<32>(c, q, s, t, sn, tn, tail); break;
        default: throw Error(SPHX_ERR_STATE, "SPHX:Ctx:fuse", "internal: fused E|A launch at this lane count");
    }
}

// One single-GPU step slot: state S[q], layout L[l].  rebuild: the step ends with re-binning into S[1-q], L[1-l]
// (7 launches); otherwise the passes write the new state straight into S[1-q] and the layout stays (5 launches).
void launch_step(sphx_ctx *c, int q, int l, int pos, bool rebuild)
{
    Clock *clk = c->clock.get();
    const int dmode = c->skin > 0.0 ? (pos == 0 ? 1 : 2) : 0;
    const FluidSet s = c->view(q, l);
    const bool track = c->skin > 0.0;
    const double *dpart = track ? (const double *)c->dpart.get() : nullptr;
    // what k_clock_scan reduces: the per-block maxima, or (many blocks) their per-tile maxima
    const double *vsrc = c->vpart.get();
    int n_red = c->n_vpart;
    const double *dpart_blocks = dpart;  // the per-block array (dpart is redirected to the tiles below)
    auto pre_reduce = [c, clk, q, dpart_blocks]() {
        if (c->n_vtiles == 0) return;
        launch(c, "k_max_tiles", k_max_tiles, dim3(c->n_vtiles), dim3(kScanBlock), (const Clock *)clk, q, c->n_vpart,
               (const double *)c->vpart.get(), dpart_blocks, c->vtile.get(), c->vtile.get() + c->n_vtiles);
    };
    if (c->n_vtiles) {
        vsrc = c->vtile.get();
        if (dpart) dpart = c->vtile.get() + c->n_vtiles;
        n_red = c->n_vtiles;
    }
    // Dual-rate loop: passes CD and E of the inner sub-steps 1 .. n_in-1 (CD of sub-step 0 comes first, E of the last
    // sub-step after).  Sub-step m reads the velocities W_m and writes W_m+1; the W alternate between the step's output
    // array and vel2 so that the last one lands in the output array.
    auto inner_substeps = [c, q, &s](FluidTmp &t) {
        if (c->n_in <= 1) { launch_physics_any(c, q, s, t, 0, 3); return; }
        double2 *const w_final = t.veln;
        auto w_of = [&](int m) { return ((c->n_in - m) & 1) ? c->vel2.get() : w_final; };  // W_m, m = 1 .. n_in
        FluidSet sm = s;
        t.veln = w_of(1);
        launch_physics_any(c, q, sm, t, 0, 3);
        for (int m = 1; m < c->n_in; ++m) {
            launch_physics_any(c, q, sm, t, 0, 4, 0, 0, 1);  // E of sub-step m-1: rho, drho and the next half-step state
            sm.vel = w_of(m);
            t.veln = w_of(m + 1);
            launch_physics_any(c, q, sm, t, 0, 3, 0, 0, 1);  // CD of sub-step m (pressure part)
        }
    };
    if (!rebuild && c->fuse_ea) {
        // pass A of this step ran inside the previous step's last launch, unless this is the first step on a fresh grid
        FluidTmp t = c->tmp_par[q], tn = c->tmp_par[1 - q];
        const FluidSet o = c->view(1 - q, l);
        t.posn = o.pos; t.veln = o.vel; t.drhon = o.drho;
        if (pos == 0) launch_physics_any(c, q, s, t, 0, 1, 1);
        launch_physics_any(c, q, s, t, 0, 2);
        inner_substeps(t);
        launch_fused_ea(c, q, s, t, o, tn);
        return;
    }
    if (!rebuild) {
        FluidTmp t = c->tmp;
        const FluidSet o = c->view(1 - q, l);
        t.posn = o.pos; t.veln = o.vel; t.drhon = o.drho;
        if (c->tail_clock) {  // 4 launches: the last workgroup of pass E advances the clock
            launch_physics_any(c, q, s, t, 0, 0, dmode, 1);
            return;
        }
        launch_physics_any(c, q, s, t, 0, 0, dmode);
        pre_reduce();
        launch(c, "k_clock_scan", k_clock_scan, dim3(1), dim3(kScanBlock), clk, q, c->phys, n_red,
               vsrc, (const double *)nullptr, (const int *)c->flags.get(), (const int *)nullptr,
               (int *)nullptr, 0, (const int *)nullptr, dpart, 0, c->half_skin(), (int *)nullptr, c->vpart_reset(), 0);
        return;
    }
    if (c->fuse_ea) {  // re-binning step: pass A came with the previous step (or stands alone at pos 0), E has a launch of its own
        FluidTmp t = c->tmp_par[q];
        if (pos == 0) launch_physics_any(c, q, s, t, 1, 1, 1);
        launch_physics_any(c, q, s, t, 1, 2);
        inner_substeps(t);
        launch_physics_any(c, q, s, t, 1, 4);
    } else {
        launch_physics_any(c, q, s, c->tmp, 1, 0, dmode);
    }
    pre_reduce();
    const FluidSet d = c->view(1 - q, 1 - l);
    if (!c->big_scan) {  // clock update and cell scan share one single-block kernel
        launch(c, "k_clock_scan", k_clock_scan, dim3(1), dim3(kScanBlock), clk, q, c->phys, n_red,
               vsrc, (const double *)nullptr, (const int *)c->flags.get(),
               (const int *)c->count.get(), d.start, c->grid.ncells, (const int *)nullptr, dpart, 1, c->half_skin(), (int *)nullptr, c->vpart_reset(), 0);
    } else {
        int *tile_sum = c->tile.get(), *tile_off = c->tile.get() + c->n_tiles + 1;
        launch(c, "k_scan_tiles", k_scan_tiles, dim3(c->n_tiles), dim3(kScanBlock), (const Clock *)clk, q,
               (const int *)c->count.get(), d.start, tile_sum, c->grid.ncells);
        launch(c, "k_clock_scan", k_clock_scan, dim3(1), dim3(kScanBlock), clk, q, c->phys, n_red,
               vsrc, (const double *)nullptr, (const int *)c->flags.get(), (const int *)tile_sum,
               tile_off, c->n_tiles, (const int *)nullptr, dpart, 1, c->half_skin(), (int *)nullptr, c->vpart_reset(), 0);
        launch(c, "k_scan_add", k_scan_add, dim3(c->n_tiles), dim3(kScanBlock), (const Clock *)clk, q, d.start,
               (const int *)tile_off, c->grid.ncells, c->n_tiles);
    }
    launch_scatter_reorder(c, clk, q, reorder_args(c->tmp.posn, c->tmp.veln, c->tmp.drhon, s.mass, s.id, d, c->tmp.src_of), d);
}

// A forced rebuild is answered with a COOL-DOWN: for the next cool_len steps every step re-bins (exactly the loop
// without a skin), then the schedule returns to K.  What outruns the skin is almost always one particle making a
// few large transport shifts in a row (max over millions of particles: at 0.5-6 M particles it happens every few
// thousand steps); shrinking K for thousands of steps -- the first policy -- cost 30 % of the throughput of long
// runs at those sizes.  The cool-down starts at 16 steps and doubles (up to 1024) while forced rebuilds keep coming
// right after it ends, so a flow that really is too fast for the skin degrades to re-binning every step.
// cool_until only changes at forced rebuilds (device-side events): the schedule stays independent of how the host
// chunks its calls.
bool slot_rebuilds(const sphx_ctx *c) { return c->pos >= c->rebuild_every - 1 || c->prov_step < c->cool_until; }

// One step slot of a dynamic context: the four passes on S[q] writing the new state into S[1-q], the clock (which
// decides rebuild_now), then the re-binning chain, every kernel of which returns at once unless rebuild_now is set:
// histogram of the new positions, scan, scatter, id-canonical reorder into temporaries, copy back in place.
void launch_step_dyn(sphx_ctx *c, int q)
{
    Clock *clk = c->clock.get();
    const FluidSet s = c->view(q, 0), o = c->view(1 - q, 0);
    FluidTmp t = c->tmp;
    t.posn = o.pos; t.veln = o.vel; t.drhon = o.drho;
    launch_physics_any(c, q, s, t, 100 + c->rebuild_every, 0, 3);
    const double *vsrc = c->vpart.get(), *dsrc = c->dpart.get();
    int n_red = c->n_vpart;
    if (c->n_vtiles) {
        launch(c, "k_max_tiles", k_max_tiles, dim3(c->n_vtiles), dim3(kScanBlock), (const Clock *)clk, q, c->n_vpart,
               (const double *)c->vpart.get(), (const double *)c->dpart.get(), c->vtile.get(), c->vtile.get() + c->n_vtiles);
        vsrc = c->vtile.get(); dsrc = c->vtile.get() + c->n_vtiles; n_red = c->n_vtiles;
    }
    launch(c, "k_clock_scan", k_clock_scan, dim3(1), dim3(kScanBlock), clk, q, c->phys, n_red, vsrc, (const double *)nullptr,
           (const int *)c->flags.get(), (const int *)nullptr, (int *)nullptr, 0, (const int *)nullptr, dsrc, 0, c->half_skin(),
           (int *)nullptr, (unsigned long long *)nullptr, c->rebuild_every);
    const int qf = q | kOnlyIfRebuild;
    const int kDynBlocks = 4096;  // grid-stride kernels: a launch that skips costs ~3 us instead of an empty 24k-block grid
    const dim3 g1(std::min(c->n_blocks_flat, kDynBlocks)), bp(kBlock);
    // Re-binning in place.  Temporaries: the tmp state arrays and the (otherwise unused) second layout's mass and id.  What
    // nobody reads during the re-ordering -- binning positions, cells, cell starts -- is written straight into the layout
    // (round 4: 52 instead of 72 bytes per particle to copy back; SPHX_DEBUG_SWITCHES=full_copyback for the old form).
    const bool lean = !debug_switches().full_copyback;
    const FluidSet d{c->posn.get(), c->veln.get(), c->drhon.get(), c->fmass_[1].get(), c->fid_[1].get(),
                     lean ? o.start : c->fstart_[1].get(), lean ? o.cell : c->fcell_[1].get(), lean ? o.posb : c->fposb_[1].get()};
    launch(c, "k_bin", k_bin, g1, bp, (const Clock *)clk, qf | kOnlyIfNoHistogram, c->grid, 0, (const double2 *)o.pos,
           c->cellid.get(), c->count.get());  // drift-triggered re-binnings only: pass E bins on the scheduled ones
    launch_cell_scan(c, clk, qf, d.start);
    launch_scatter_reorder(c, clk, qf, reorder_args(o.pos, o.vel, o.drho, s.mass, s.id, d, c->tmp.src_of), d, kDynBlocks);
    CopyBack cb{d.pos, d.vel, d.posb, o.pos, o.vel, o.posb, d.drho, d.mass, o.drho, o.mass, d.id, d.cell, d.start,
                o.id, o.cell, o.start, c->grid.ncells + 1, lean ? 1 : 0};
    launch(c, "k_copyback", k_copyback, g1, bp, (const Clock *)clk, qf, cb);
}

// host-side bookkeeping of one executed step
void track_step(sphx_ctx *c)
{
    if (c->dyn) {
        c->prov_step += 1;
        c->cur ^= 1;
        return;
    }
    const bool rebuild = slot_rebuilds(c);
    c->prov_step += 1;
    c->out_lay = c->lay;  // outputs are stored in the layout the step ran in; after a rebuild tmp.src_of maps to it
    c->out_par = c->cur;  // ... and (fuse_ea) in the record buffers of the step's state parity
    c->cur ^= 1;
    if (rebuild) { c->lay ^= 1; c->pos = 0; c->n_rebins += 1; }
    else c->pos += 1;
}

// arm the device clock for a batch (k_prepare) and count it (Clock::seq / sphx_ctx::host_seq)
void arm_clock(sphx_ctx *c, double t_target, long long max_steps, int q0, const double *vmax_in)
{
    c->host_seq += 1;
    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(1), 0, c->stream, c->clock.get(), c->phys, t_target, max_steps, q0, vmax_in);
}

int graph_slots(const sphx_ctx *c)
{
    const int period = c->dyn ? 2 : 2 * c->rebuild_every;  // (cur, lay, pos) returns to itself after 2K steps
    return period * std::max(1, c->spg / period);
}

constexpr int kMinGraphSlots = 4;    // shorter exact batches are launched eagerly (a replay costs ~50 us of host time)
constexpr size_t kMaxGraphs = 96;    // cache bound: 2 * 2 * K phases for a caller with one cadence, K <= 16 in practice

// the graph of `n` step slots entered at phase (cur, lay, pos) of the static schedule (dynamic contexts: only cur matters)
sphx_ctx::CachedGraph &get_graph(sphx_ctx *c, int cur, int lay, int pos, int n)
{
    const std::array<int, 4> key{cur, c->dyn ? 0 : lay, c->dyn ? 0 : pos, n};
    auto it = c->graphs.find(key);
    if (it != c->graphs.end()) return it->second;
    if (c->graphs.size() >= kMaxGraphs) {
        SPHX_HIP(hipStreamSynchronize(c->stream));  // earlier batches may still be replaying the graphs about to be destroyed
        c->drop_graph();
    }
    const int K = c->rebuild_every;
    const bool prof = c->profiling;
    c->profiling = false;
    SPHX_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    try {
        int q = cur, l = lay, p = pos;
        for (int j = 0; j < n; ++j) {
            if (c->dyn) {
                launch_step_dyn(c, q);
            } else {
                const bool rebuild = p == K - 1;
                launch_step(c, q, l, p, rebuild);
                if (rebuild) { l ^= 1; p = 0; }
                else ++p;
            }
            q ^= 1;
        }
    } catch (...) {
        hipGraph_t junk = nullptr;
        (void)hipStreamEndCapture(c->stream, &junk);
        if (junk) (void)hipGraphDestroy(junk);
        c->profiling = prof;
        throw;
    }
    c->profiling = prof;
    hipGraph_t g = nullptr;
    SPHX_HIP(hipStreamEndCapture(c->stream, &g));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    SPHX_HIP(e);
    sphx_ctx::CachedGraph &g_new = c->graphs[key];
    g_new.exec = exec;
    c->graphs_captured += 1;
    return g_new;
}

// Enqueue `slots` step slots from the current (cur, lay, pos); slots that find run[q]==0 are no-ops.  Whole graphs
// are replayed from ANY phase of the static schedule (see sphx_ctx::graphs); what is left over is launched eagerly,
// or -- exact_tail: the caller asked for exactly this many steps -- replayed as a graph of its own.  Only a cool-down
// (every slot re-bins, see slot_rebuilds) runs eagerly throughout.  The host copies of cur/lay/pos advance as if
// every slot executed; read_clock() recomputes them from the executed step count.
// capture_only: build the graphs this call would replay, launch nothing (sphx_ctx_prepare_steps).
void enqueue_slots(sphx_ctx *c, int64_t slots, bool exact_tail, bool capture_only = false)
{
    int64_t left = slots;
    // capture_only walks the phases on copies
    const int cur0 = c->cur, lay0 = c->lay, pos0 = c->pos, out_lay0 = c->out_lay;
    const int64_t prov0 = c->prov_step;
    while (left > 0) {
        const int per_graph = graph_slots(c);
        const bool steady = c->prov_step >= c->cool_until;
        int n = 0;
        if (!c->profiling && steady) {
            if (left >= per_graph) n = per_graph;
            else if (exact_tail && left >= kMinGraphSlots) n = (int)left;
        }
        if (n > 0) {
            sphx_ctx::CachedGraph &cg = get_graph(c, c->cur, c->lay, c->pos, n);
            if (capture_only) {
                // First replays are slow (~4 us per step slot on ROCm 7.2: the executable graph is set up on the device
                // at its first launch, hipGraphUpload does not take that over): a graph that has never run is replayed
                // once now with the clock disarmed (k_disarm, see sphx_ctx_prepare_steps), so every kernel of every slot
                // returns at once and nothing changes.  Small channels only: there a step is ~20 us; on millions of
                // particles a first replay is noise and an idle pass over the arrays is not free.
                if (!cg.launched && c->n_blocks_particles <= 8192) {
                    SPHX_HIP(hipGraphLaunch(cg.exec, c->stream));
                    cg.launched = true;
                }
            } else {
                SPHX_HIP(hipGraphLaunch(cg.exec, c->stream));
                cg.launched = true;
                c->slots_replayed += n;
            }
            for (int k = 0; k < n; ++k) track_step(c);
            left -= n;
            continue;
        }
        if (!capture_only) {
            if (c->dyn) launch_step_dyn(c, c->cur);
            else launch_step(c, c->cur, c->lay, c->pos, slot_rebuilds(c));
            c->slots_eager += 1;
        }
        track_step(c);
        --left;
    }
    if (capture_only) {
        c->cur = cur0; c->lay = lay0; c->pos = pos0; c->out_lay = out_lay0; c->prov_step = prov0;
    }
    SPHX_HIP(hipGetLastError());
}

void set_epoch(sphx_ctx *c)
{
    c->epoch_step = c->h_clock->step;
    c->epoch_cur = c->cur; c->epoch_lay = c->lay; c->epoch_pos = c->pos; c->epoch_out_lay = c->out_lay;
    c->epoch_out_par = c->out_par;
    c->epoch_n_rebins = c->n_rebins;
    c->prov_step = c->epoch_step;
}

// wait for the stream: poll for a while (a blocking wait costs ~10 us of wake-up latency, which a 20-step batch of a
// small channel notices), then block
void wait_stream(sphx_ctx *c)
{
    for (int k = 0; k < 20000; ++k) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) SPHX_HIP(e);
        (void)hipGetLastError();
    }
    SPHX_HIP(hipStreamSynchronize(c->stream));
}

void read_clock(sphx_ctx *c)
{
    wait_stream(c);
    if (c->h_pub && c->h_pub->seq == c->host_seq) {
        *c->h_clock = *c->h_pub;  // the device published the clock when the last batch stopped
    } else {  // nothing armed since the last read, or the batch ran out of slots with the loop still going
        SPHX_HIP(hipMemcpyAsync(c->h_clock, c->clock.get(), sizeof(Clock), hipMemcpyDeviceToHost, c->stream));
        SPHX_HIP(hipStreamSynchronize(c->stream));
    }
    c->timer.collect();
    // replay the bookkeeping of the steps that really executed since the last read
    const int64_t executed = (int64_t)c->h_clock->step - c->epoch_step;
    c->cur = c->epoch_cur; c->lay = c->epoch_lay; c->pos = c->epoch_pos; c->out_lay = c->epoch_out_lay;
    c->out_par = c->epoch_out_par;
    c->n_rebins = c->epoch_n_rebins;
    c->prov_step = c->epoch_step;
    if (executed > 0) {
        int64_t left = executed;
        if (c->dyn) {  // only the parity matters
            c->prov_step += left - (left & 1);
            left &= 1;
        }
        while (left > 0) {
            if (c->prov_step >= c->cool_until && left > 4 * c->rebuild_every) {
                // steady interval: (cur, lay, pos) repeats every 2K steps -> skip whole periods
                const int64_t period = 2 * c->rebuild_every, skip = ((left - 1) / period - 1) * period;
                if (skip > 0) { c->prov_step += skip; left -= skip; c->n_rebins += 2 * (skip / period); }
            }
            track_step(c);
            --left;
        }
        c->have_step_outputs = true;
    }
    set_epoch(c);
}

// Re-bin the current state into the other buffers without taking a step: the device stopped the loop because
// some particle drifted further than skin/2 from where it was binned.  Happens only when the flow is faster than
// the skin was sized for; the rebuild interval shrinks by one when it happens in back-to-back cycles.
void forced_rebuild(sphx_ctx *c)
{
    const int q = c->cur, l = c->lay;
    if (c->have_step_outputs && c->out_lay != l)
        throw Error(SPHX_ERR_STATE, "SPHX:Ctx:rebuild", "internal: forced rebuild straight after a rebuild step");
    const FluidSet s = c->view(q, l), d = c->view(1 - q, 1 - l);
    hipStream_t st = c->stream;
    const int n = c->h_clock->n;
    const dim3 g1(c->n_blocks_flat), bp(kBlock);
    const bool prof = c->profiling;
    c->profiling = false;
    hipLaunchKernelGGL(k_bin, g1, bp, 0, st, (const Clock *)nullptr, 0, c->grid, n, (const double2 *)s.pos, c->cellid.get(),
                       c->count.get());
    launch_cell_scan(c, nullptr, 0, d.start);
    hipLaunchKernelGGL(k_scatter, g1, bp, 0, st, (const Clock *)nullptr, 0, n, (const int *)c->cellid.get(), c->count.get(),
                       (const int *)d.start, c->perm.get());
    // src_of: new slot -> slot of the layout the last step's outputs (rho, p, force, Vol, B) are stored in
    hipLaunchKernelGGL(k_reorder, g1, bp, 0, st, (const Clock *)nullptr, 0, n, (const int *)c->cellid.get(),
                       (const int *)d.start, (const int *)c->perm.get(),
                       reorder_args(s.pos, s.vel, s.drho, s.mass, s.id, d, c->tmp.src_of));
    hipLaunchKernelGGL(k_rebinned, dim3(1), dim3(1), 0, st, c->clock.get());
    c->profiling = prof;
    SPHX_HIP(hipGetLastError());
    c->cur = 1 - q; c->lay = 1 - l; c->pos = 0;  // out_lay stays l
    c->h_clock->need_rebuild = 0;
    c->h_clock->drift = 0.0;
    if (c->h_pub) { c->h_pub->need_rebuild = 0; c->h_pub->drift = 0.0; }  // (the published copy may be read again before the next batch)
    // cool-down (see slot_rebuilds): 16 steps, doubled while the next forced rebuild follows the previous cool-down
    // within two rebuild cycles
    const int64_t now = c->h_clock->step;
    const bool again = c->n_forced_rebuilds > 0 && now - c->cool_until <= 2 * (int64_t)c->rebuild_every;
    c->cool_len = again ? std::min<int64_t>(2 * std::max<int64_t>(c->cool_len, 16), 1024) : 16;
    c->cool_until = now + c->cool_len;
    if (debug_switches().log)
        fprintf(stderr, "sphx: forced rebuild #%lld at step %lld (drift bound hit): re-binning every step for %lld steps\n",
                (long long)c->n_forced_rebuilds + 1, (long long)now, (long long)c->cool_len);
    c->last_forced_step = now;
    c->n_forced_rebuilds += 1;
    set_epoch(c);
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
    // enough lanes to put ~4 waves on each of the 1024 SIMDs; 4..32 lanes per particle (measured: with the
    // balanced neighbour list 4 lanes beat 1-2 even at 6 M particles, 32 beat 16 at 5 k)
    // Round 2 (entries ahead, LDS tile in the force pass): 2 lanes beat 4 from 0.5 M particles (189 vs 193 us/step) and
    // clearly at 6 M (2195 vs 2416); at 65 k particles 4 lanes keep more waves in flight (51 vs 60 us/step).
    // With pass E and the next pass A in one launch and the clock in its tail (up to 2048 workgroups: 131 k particles at
    // 4 lanes, 262 k at 2) the three-launch step of 2 lanes beats the five launches of 4: 194 k particles 90.9 vs 96.4 us/step,
    // 259 k 109.1 vs 117.4; at 130 k, where both fuse, 4 lanes win (68.0 vs 70.6).
    // Round 3 (the fused launch now reaches 4 096 workgroups = 262 k particles at 4 lanes): both lane counts fused, 4 lanes win
    // at 151 k (76.4 vs 79.8) and 194 k (89.9 vs 90.9), 2 lanes from 259 k (110.2 vs 110.8) -> the switch moves to 220 k.
    const long target = 256L * 4 * 4 * 64;
    if (nf > 220000) return 2;
    int lpp = 4;
    while (lpp < 32 && (long)nf * lpp * 2 <= target) lpp *= 2;
    // Round 3, compact kernels with passes E and A in one launch: 32 lanes win at 1 875 fluid particles (16.1 vs 17.2 us/step),
    // tie at 2 494, and lose from 3 300 (18.1 vs 17.5; C2 = 4 800: 19.9 vs 19.3; 5 852: 20.9 vs 19.7) -- profiles/r03_lanes_small.txt
    if (lpp == 32 && nf > 2400) lpp = 16;
    return lpp;
}

// rows of the per-lane neighbour list: >= 96 entries per particle (fluid ring ~30-45 + wall ring up to ~20)
int nl_cap_for(int lpp) { return std::max(96 / lpp, 16); }

// sort `n` particles given in arbitrary order into cell order on the device (context creation)
void initial_sort(sphx_ctx *c, const Grid &g, int n, const double2 *pos, int *cellid, int *count, int *start, int *perm,
                  const ReorderArgs &ra)
{
    hipStream_t s = c->stream;
    if (n <= 0) {
        SPHX_HIP(hipMemsetAsync(start, 0, ((size_t)g.ncells + 1) * sizeof(int), s));
        return;
    }
    hipLaunchKernelGGL(k_bin, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, (const Clock *)nullptr, 0, g, n, pos, cellid, count);
    hipLaunchKernelGGL(k_scan_only, dim3(1), dim3(kScanBlock), 0, s, (const Clock *)nullptr, 0, (const int *)count, start, g.ncells);
    hipLaunchKernelGGL(k_scatter, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, (const Clock *)nullptr, 0, n,
                       (const int *)cellid, count, (const int *)start, perm);
    hipLaunchKernelGGL(k_reorder, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, (const Clock *)nullptr, 0, n,
                       (const int *)cellid, (const int *)start, (const int *)perm, ra);
    SPHX_HIP(hipGetLastError());
}

Phys make_phys(const sphx_params *prm)
{
    Phys ph{};
    ph.kc = make_kernel_const(prm->h);
    ph.rho0 = prm->rho0; ph.inv_sigma0 = prm->inv_sigma0; ph.mu = prm->mu; ph.p0 = prm->p0; ph.c_f = prm->c_f;
    ph.g = prm->gravity_g; ph.tc = prm->transport_coeff; ph.nu = prm->mu / prm->rho0; ph.DL = prm->DL; ph.DH = prm->DH;
    ph.w0 = ph.kc.sigma;
    ph.dt_viscous = 0.125 * prm->h * prm->h / std::max(ph.nu, 1e-12);
    ph.dt_body = 0.25 * std::sqrt(prm->h / std::max(std::fabs(ph.g), 1e-12));
    return ph;
}

// allocate every device array for `cap` particles and the grid of the context
void ctx_alloc(sphx_ctx *c, int cap)
{
    const Grid &g = c->grid;
    c->cap = cap;
    c->n_blocks_particles = (int)div_up((size_t)cap * c->lpp, kBlock);
    c->n_blocks_flat = (int)div_up((size_t)cap, kBlock);
    for (int k = 0; k < 2; ++k) {
        c->fpos_[k].alloc(cap); c->fvel_[k].alloc(cap); c->fdrho_[k].alloc(cap); c->fmass_[k].alloc(cap);
        c->fid_[k].alloc(cap); c->fstart_[k].alloc((size_t)g.ncells + 1); c->fcell_[k].alloc(cap);
        if (c->skin > 0.0) c->fposb_[k].alloc(cap);
    }
    c->posn.alloc(cap); c->veln.alloc(cap); c->ffp.alloc(cap); c->ff.alloc(cap); c->fa.alloc(cap); c->fB.alloc(cap);
    c->drhon.alloc(cap); c->rho_out.alloc(cap); c->p_out.alloc(cap); c->fvol.alloc(cap); c->fvol.zero(c->stream);
    c->posn.zero(c->stream); c->veln.zero(c->stream); c->ffp.zero(c->stream); c->ff.zero(c->stream);
    c->fa.zero(c->stream); c->fB.zero(c->stream); c->drhon.zero(c->stream); c->rho_out.zero(c->stream); c->p_out.zero(c->stream);
    c->n_vpart = c->n_blocks_particles;
    c->vpart.alloc(c->n_vpart);
    // Small channels with a skin: move steps are 4 launches, the clock update rides in pass E (continuity_tail);
    // vpart entries then double as "ready" flags and start out empty (all ones)
    c->tail_clock = c->skin > 0.0 && !c->is_slab && !c->dyn && c->n_vpart <= tail_clock_limit() && !debug_switches().no_tail_clock;
    // (skinned slabs: the same hand-over feeds slab_seal_tail)
    const bool vpart_flags = c->tail_clock || (c->is_slab && c->rebuild_every > 1);
    SPHX_HIP(hipMemsetAsync(c->vpart.get(), vpart_flags ? 0xFF : 0, (size_t)c->n_vpart * sizeof(double), c->stream));
    c->dpart.alloc(c->n_vpart);
    c->dpart.zero(c->stream);
    c->n_vtiles = (!c->is_slab && c->n_vpart > 4 * kMaxTile) ? (int)div_up((size_t)c->n_vpart, kMaxTile) : 0;
    if (c->n_vtiles) c->vtile.alloc(2 * (size_t)c->n_vtiles);
    c->cellid.alloc(cap); c->count.alloc((size_t)g.ncells + 1); c->perm.alloc(cap); c->src_of.alloc(cap);
    c->count.zero(c->stream);
    const int nl_cap = nl_cap_for(c->lpp);
    const size_t stride = (size_t)c->n_blocks_particles * kBlock;  // one list column per launched lane
    c->nl_idx.alloc(stride * nl_cap);
    c->nl_cnt.alloc(stride);
    c->nl_cnt.zero(c->stream);
    const size_t nl_words = (size_t)(nl_cap + 1) / 2;  // two 16-bit rows per word
    if (c->walk_kernels) { c->nl_pk.alloc(stride * std::max<size_t>(nl_words, 2)); c->nl_pk.zero(c->stream); }
    const int sl_cap = c->skin > 0.0 ? (3 * nl_cap + 1) / 2 : 0;
    if (sl_cap) {
        c->sl_idx.alloc(stride * sl_cap);
        c->sl_cnt.alloc(stride);
        c->sl_cnt.zero(c->stream);
        if (c->walk_kernels) { c->sl_pk.alloc(stride * std::max<size_t>((size_t)(sl_cap + 1) / 2, 4)); c->sl_pk.zero(c->stream); }
    }
    const double sl_r = 2.0 * c->prm.h + c->skin;
    c->flags.alloc(1);
    c->flags.zero(c->stream);
    c->big_scan = g.ncells > kBigScanCells;
    c->n_tiles = (int)div_up((size_t)g.ncells, kScanBlock);
    c->tile.alloc(2 * ((size_t)c->n_tiles + 1));
    c->tmp = FluidTmp{c->posn.get(), c->veln.get(), c->drhon.get(), c->fa.get(), c->fB.get(), c->ffp.get(), c->ff.get(),
                      c->rho_out.get(), c->p_out.get(), c->cellid.get(), c->count.get(), c->perm.get(), c->src_of.get(),
                      c->vpart.get(), c->dpart.get(), c->nl_idx.get(), c->nl_cnt.get(), c->flags.get(), c->tile.get(),
                      (int)stride, nl_cap, c->sl_idx.get(), c->sl_cnt.get(), sl_cap, sl_r * sl_r, cap, c->n_vpart,
                      c->skin > 0.0 ? c->half_skin() : -1.0, c->fvol.get(), c->nl_pk.get(), c->sl_pk.get(), c->is_slab ? 1 : 0, nullptr, 0, nullptr};
    // (2 = never: a slab hands out its state only, sphx_slab_snapshot; the dual-rate loop, which reads force_prior in its
    //  inner sub-steps, runs on the compact kernels only)
    c->tmp.lazy_out = (c->walk_kernels && !debug_switches().no_lazy_out) ? (c->is_slab ? 2 : 1) : 0;
    if (c->walk_kernels) {  // (zeros = empty layouts until the first cell sweep has run)
        c->tmap.alloc(8 * (size_t)c->n_blocks_particles); c->tmap.zero(c->stream);
        c->tmp.tmap = c->tmap.get();
    }
    // E|A fusion: small static-schedule channels on the compact kernels (the clock rides in the tail workgroup)
    c->fuse_ea = c->tail_clock && (c->lpp >= 16 || (c->walk_kernels && !c->lds_tiles_be && c->lpp >= 2)) && !debug_switches().no_fuse_ea;
    c->tmp_par[0] = c->tmp;
    c->tmp_par[1] = c->tmp;
    if (c->fuse_ea) {
        c->fa2.alloc(cap); c->fvol2.alloc(cap); c->fa2.zero(c->stream); c->fvol2.zero(c->stream);
        c->nl_idx2.alloc(stride * nl_cap); c->nl_cnt2.alloc(stride); c->nl_cnt2.zero(c->stream);
        c->tmp_par[1].a = c->fa2.get(); c->tmp_par[1].vol = c->fvol2.get();
        c->tmp_par[1].nl_idx = c->nl_idx2.get(); c->tmp_par[1].nl_cnt = c->nl_cnt2.get();
        if (c->walk_kernels) {
            c->nl_pk2.alloc(stride * std::max<size_t>(nl_words, 2)); c->nl_pk2.zero(c->stream);
            c->tmp_par[1].nl_pk = c->nl_pk2.get();
        }
    }
    // Opt-in dual-rate loop (sphx_params::dual_rate): small channels on the compact kernels, where the fused launches give
    // the step a fixed shape.  The number of inner sub-steps is fixed per context (the graphs are static): how many acoustic
    // steps fit into the viscous / body-force step, at most dual_rate.  Fine channels are viscous-limited: n_in = 1 there.
    c->n_in = (c->fuse_ea && c->lpp >= 16) ? dual_rate_substeps(c->prm) : 1;
    if (c->n_in > 1) { c->vel2.alloc(cap); c->vel2.zero(c->stream); }

    c->tau_part.alloc((size_t)2 * c->n_blocks_flat);
    c->tau_out.alloc(2);
}

// upload n fluid particles (host SoA) and sort them into state 0 / layout 0
void upload_fluid(sphx_ctx *c, int n, const double *hx, const double *hy, const double *hvx, const double *hvy,
                  const double *hdrho, const double *hmass, const int *hid, bool wrap)
{
    hipStream_t s = c->stream;
    std::vector<double2> hp((size_t)std::max(n, 1)), hv((size_t)std::max(n, 1));
    for (int i = 0; i < n; ++i) { hp[i] = make_double2(hx[i], hy[i]); hv[i] = make_double2(hvx[i], hvy[i]); }
    if (n > 0) {
        c->posn.upload(hp.data(), n, s); c->veln.upload(hv.data(), n, s);
        c->drhon.upload(hdrho, n, s);
        c->fmass_[1].upload(hmass, n, s);
        if (hid) c->fid_[1].upload(hid, n, s);
        else hipLaunchKernelGGL(k_iota, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, n, c->fid_[1].get(), 0);
        if (wrap) hipLaunchKernelGGL(k_wrap_x, dim3(div_up(n, kBlock)), dim3(kBlock), 0, s, n, c->posn.get(), c->prm.DL);
    }
    const FluidSet d = c->view(0, 0);
    initial_sort(c, c->grid, n, c->posn.get(), c->cellid.get(), c->count.get(), d.start, c->perm.get(),
                 reorder_args(c->posn.get(), c->veln.get(), c->drhon.get(), c->fmass_[1].get(), c->fid_[1].get(), d, nullptr));
    SPHX_HIP(hipStreamSynchronize(s));  // host staging vectors die after this
}

// upload nw wall particles (host SoA, x already in this context's frame) and sort them once
void upload_walls(sphx_ctx *c, int nw, const double *hx, const double *hy, const double *hmass, const double *hvx,
                  const double *hvy, const int *hid, bool wrap)
{
    const Grid &g = c->grid;
    hipStream_t s = c->stream;
    const size_t nwz = nw > 0 ? (size_t)nw : 1;
    c->wpos.alloc(nwz); c->wa.alloc(nwz); c->wid.alloc(nwz);
    c->wstart.alloc((size_t)g.ncells + 1); c->wrow_any.alloc(g.ncy);
    DevBuf<double2> tpos(nwz);
    DevBuf<double4> ta(nwz);
    DevBuf<int> tid(nwz), tcell(nwz), tperm(nwz), tcount((size_t)g.ncells + 1);
    tcount.zero(s);
    std::vector<double2> hp(nwz);
    std::vector<double4> ha(nwz);
    for (int i = 0; i < nw; ++i) {
        hp[i] = make_double2(hx[i], hy[i]);
        ha[i] = make_double4(hmass[i] / c->prm.rho0, hvx[i], hvy[i], 0.0);  // walls keep rho = rho0 (sph_physics_mex.c:214-216,233)
    }
    if (nw > 0) {
        tpos.upload(hp.data(), nw, s); ta.upload(ha.data(), nw, s); tid.upload(hid, nw, s);
        if (wrap) hipLaunchKernelGGL(k_wrap_x, dim3(div_up(nw, kBlock)), dim3(kBlock), 0, s, nw, tpos.get(), c->prm.DL);
    }
    ReorderArgs ra{};
    ra.n2 = 1; ra.src2[0] = tpos.get(); ra.dst2[0] = c->wpos.get();
    ra.n4 = 1; ra.src4[0] = ta.get(); ra.dst4[0] = c->wa.get();
    ra.id_src = tid.get(); ra.id_dst = c->wid.get(); ra.src_of = nullptr; ra.cell_dst = nullptr;
    initial_sort(c, g, nw, tpos.get(), tcell.get(), tcount.get(), c->wstart.get(), tperm.get(), ra);
    hipLaunchKernelGGL(k_row_any, dim3(div_up(g.ncy, 64)), dim3(64), 0, s, g, (const int *)c->wstart.get(), c->wrow_any.get());
    SPHX_HIP(hipGetLastError());
    SPHX_HIP(hipStreamSynchronize(s));  // temporaries die here
    c->walls = Walls{c->wpos.get(), c->wa.get(), c->wid.get(), c->wstart.get(), c->wrow_any.get(), nw};
}

void init_clock(sphx_ctx *c, int n, double t0, int64_t step0)
{
    hipStream_t s = c->stream;
    c->clock.alloc(1);
    Clock k{};
    k.t = t0; k.dt = 0.0; k.dt_last = 0.0; k.t_target = t0; k.t_end = c->prm.t_end; k.vmax = 0.0;
    k.step = step0; k.steps_left = -1; k.run[0] = 0; k.run[1] = 0; k.status = 0; k.n = n;
    k.drift = 0.0; k.need_rebuild = 0;
    k.fresh = 1; k.rebuild_now = 0; k.pos_count = 0; k.n_drift_rebuilds = 0;
    k.seq = 0;
    k.n_in = c->n_in;
    if (!c->h_pub) SPHX_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_pub), sizeof(Clock), hipHostMallocMapped));
    void *pub_dev = nullptr;
    SPHX_HIP(hipHostGetDevicePointer(&pub_dev, c->h_pub, 0));
    k.pub = static_cast<Clock *>(pub_dev);
    *c->h_pub = k;
    c->h_pub->seq = -1;  // nothing published yet
    c->host_seq = 0;
    *c->h_clock = k;
    SPHX_HIP(hipMemcpyAsync(c->clock.get(), c->h_clock, sizeof(Clock), hipMemcpyHostToDevice, s));
    SPHX_HIP(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_vmax_init, dim3(1), dim3(kScanBlock), 0, s, c->clock.get(), c->grid, (const double2 *)c->fpos_[0].get(),
                       (const double2 *)c->fvel_[0].get(), (double *)nullptr);
    SPHX_HIP(hipGetLastError());
    c->cur = 0; c->lay = 0; c->pos = 0; c->out_lay = 0;
    set_epoch(c);
    c->pending_target = step0;
}

void common_checks(const sphx_params *prm, int n_fluid, int n_total)
{
    require(prm != nullptr, "SPHX:Ctx:params", "params must not be NULL");
    require(prm->dual_rate >= 0 && prm->dual_rate <= 4, "SPHX:Ctx:dual_rate",
            "dual_rate must be 0, 1 or 2..4 (inner sub-steps per outer step)");
    require(n_total > 0 && n_fluid > 0 && n_fluid <= n_total, "SPH:Neighbor:count",
            "Invalid n_fluid/n_total or inconsistent pos size.");
    require(prm->h > 0.0 && prm->DL > 0.0, "SPH:Neighbor:param", "h and DL must be positive.");
}

void y_extent(const double *py, int n_total, double &y_min, double &y_max)
{
    y_min = py[0]; y_max = py[0];
    for (int i = 1; i < n_total; ++i) { y_min = std::min(y_min, py[i]); y_max = std::max(y_max, py[i]); }
    require(std::isfinite(y_min) && std::isfinite(y_max), "SPH:Neighbor:pos", "pos must be finite.");
}

// Which kernel forms the context runs, once lanes per particle and the grid are known.  Up to 8 lanes per particle: the
// large-channel ("_w") forms, whose lists hold fluid neighbours as 16-bit index differences -- a neighbour sits at most one
// cell column (+ a few cells) away, so a column must hold well under 2^15 particles; a grid with taller columns (a channel
// thousands of particles high) goes to 16 lanes per particle and the compact kernels instead.
void choose_kernel_forms(sphx_ctx *c, bool lpp_given, double column_load, int n_resident)
{
    if (c->lpp <= 8 && 1.3 * column_load + 64.0 > (double)kDeltaMax) {
        if (lpp_given)
            throw Error(SPHX_ERR_ARG, "SPHX:Ctx:lpp", "cell columns too populous for the 16-bit neighbour lists of the "
                                                      "large-channel kernels: use lanes_per_particle >= 16 (or 0 = auto)");
        c->lpp = 16;
    }
    const DebugSwitches &dbg = debug_switches();
    c->walk_kernels = c->lpp <= 8;
    c->sweep_kernels = c->walk_kernels;
    c->lds_tiles = c->walk_kernels && !dbg.no_lds_tiles;  // (65 k particles: 46.8 with, 46.3 us/step without;
    // 100-130 k: equal; 194 k: 96.4 / 97.1; 259 k: 117.3 / 119.5; 360 k: 145.8 / 151.0 -- kept on at every size)
    // (round 3, layouts stored + slot-coded lists: 0.5 M 164 without / 175 us/step with -- it gives up the fused E|A launch --,
    //  0.83 M 260 / 260, 1.3 M 430 / 420; before that: 270 / 285 at 0.83 M, 440 / 457 at 1.3 M, hence 2 M then)
    //  A slab has no fused launch to give up: in-process rings of 0.51 M / 0.76 M-particle slabs run 5 % / 9 % faster with the
    //  tiles (C5 as 12 / 8 slabs: 2 970 -> 2 821, 2 546 -> 2 323 us/step), 0.25 M-particle slabs do not (276 / 279); a single
    //  context of 0.76 M particles is a tie (239.5 / 239.8).
    //  Round 4 (walks that do not look at the entry, two-slot staging): a single context takes the tiles from where the fused
    //  E|A launch ends (4096 workgroups per pass = 524 k particles at 2 lanes each) -- 0.59 M particles 189 / 198 -> 181 / 186
    //  us/step (window / sustained), 0.79 M 242 / 257 -> 231 / 240; below, the fused launch is worth more: 0.52 M 165 / 172
    //  fused against 170 / 176 with tiles (profiles/r04_tiles_from_*.txt).  It was 10^6 before.
    //  Slabs: from 0.3 M particles held (0.5 M before): C5 as 16 slabs of 0.39 M 3 539 -> 3 221 us/step in an in-process ring,
    //  slabs of 0.24 M 290 / 290 (profiles/r04_slab_tiles_from.txt).
    const bool fits_fused = div_up((size_t)std::max(n_resident, 0) * (size_t)c->lpp, (size_t)kBlock) <= (size_t)tail_clock_limit();
    const int tiles_from = dbg.tiles_be_from > 0 ? dbg.tiles_be_from : (c->is_slab ? 300000 : (fits_fused ? 1000000 : 0));
    c->lds_tiles_be = c->lds_tiles && c->lpp <= 2 && n_resident >= tiles_from;
    c->lds_tiles_a = c->lds_tiles_be;  // (6 M particles: 450 with, 478 us without; 0.5 M: 50.5 with, 42.3 without)
    // every pass stages the same layout: the lists can name its slots (kSlotCodes) -- the index differences that remain need a
    // little more room than the plain ones
    static_assert(tile_slots(2) == kSlotCodes, "the layout the lists are coded against is the tile of passes A, B and E");
    c->coded_lists = c->lds_tiles_be && c->lpp == 2 && 1.3 * column_load + 64.0 <= (double)kCodedDeltaMax && !dbg.no_coded_lists;
}

void check_lpp(int lpp)
{
    require(lpp == 1 || lpp == 2 || lpp == 4 || lpp == 8 || lpp == 16 || lpp == 32, "SPHX:Ctx:lpp",
            "lanes_per_particle must be 1,2,4,8,16 or 32");
}

void ctx_setup(sphx_ctx *c, const sphx_params *prm, int n_fluid, int n_total, const double *pos, const double *vel,
               const double *drho_dt, const double *mass, const double *wall_vel, double t0, int64_t step0)
{
    common_checks(prm, n_fluid, n_total);
    ensure_device();
    c->prm = *prm;
    c->nf = n_fluid;
    c->nt = n_total;
    c->nw = n_total - n_fluid;
    const int nf = c->nf, nw = c->nw;
    const size_t ntz = (size_t)n_total;
    const double *px = pos, *py = pos + ntz;

    c->lpp = prm->lanes_per_particle > 0 ? prm->lanes_per_particle : pick_lpp(nf);
    check_lpp(c->lpp);
    c->spg = prm->steps_per_graph > 0 ? prm->steps_per_graph : 40;  // measured at C2: 10 -> 25.4, 40 -> 25.1, 80 -> 24.8 us/step
    if (c->spg & 1) c->spg += 1;

    // Rebuild interval K and the cell skin that pays for it.  One step moves a particle by at most
    // dt*v <= 0.25 h v/(c_f+v); the reference sets c_f = 10 U_max, so v/c_f stays near 0.1-0.15 (0.033 h per
    // step; measured 0.017 h at c_f = 15) plus transport shifts of single particles (up to ~0.09 h measured):
    // the skin covers K-1 steps of 0.035 h.  If a flow outruns it the device clock stops the loop and
    // forced_rebuild() takes over, so this is a speed heuristic, never a correctness assumption.
    // Measured (MI355X, us/step, K=1 -> K=5): 4.8 k particles 47.5 -> 37.5 (launch-bound: 7 -> 5 launches per
    // step), 60 k 93.9 -> 80.2, 0.5 M 343 -> 316, 6 M 4101 -> 3907 (the wider cells lengthen the candidate sweeps
    // by ~10 %, the scatter/reorder kernels run on every fifth step only).
    // With the superset list: K = 5 / 8 / 10 give 24.9 / 24.3 / 24.1 us/step at 5 k particles (fewer rebuild
    // launches per step) but 2443 / 2490 / 2488 at 6 M (longer superset lists) -> 8 for small channels, 5 otherwise.
    // Round 2 (pass E and the next pass A share a launch: a step is 3 launches, a re-binning step 6): 5 k particles,
    // K = 8 / 12 / 16 / 24 / 32 -> 21.1 / 19.95 / 19.6 / 19.6 / 19.4 us/step, the 20 s run's L2 0.85 / 0.85 / 0.80 / 0.90 % -> 16.
    // 20 k - 250 k particles, K = 5 / 8 / 12 / 16: 28.8 / 27.6 / 27.1 / 27.8 us/step at 21 k, 49.8 / 48.7 / 49.1 / 49.4 at 65 k,
    // 79.5 / 72.1 / 72.6 / 74.8 at 130 k, 142 / 118 / 119 / 121 at 250 k (K = 5: its thin skin forces rebuilds + cool-downs) -> 8;
    // 0.5 M: 189 / 192 at K = 5 / 8, 6 M: 2 201 / 2 262 -> 5.
    // Round 4: the interval and the skin are two things (profiles/r04_k_skin_*.txt).  The skin prices EVERY step (the superset
    // list grows with it: 1.30 x the neighbours at 0.28 h, 1.55 x at 0.49 h) and, with the drift check, is all that correctness
    // needs; K only says when a re-binning is taken whether or not the drift asks for one.  Where the device re-bins by itself
    // (from 10^6 particles) a late scheduled re-binning costs nothing -- the drift bound comes first -- so the skin stays that
    // of K = 5 and K goes up: C5 1 632 -> 1 507 us/step in bench.py's window at K = 12, 1 723 -> 1 598 sustained (K = 12 with
    // 0.31 h: 1 514 / 1 603); on another box K = 12 / 16 / 24 / 64: 1 550 / 1 523 / 1 503 / 1 493 in the window, 1 645 sustained for
    // all four (the drift bound decides there) -> 24.  1.25 M particles 384 -> 354 / 413 -> 401.  On the host-driven schedule a
    // drift stop is a round trip and a cool-down, so the skin has to grow with K: 0.3-2 M particles K = 10 with 0.42 h (C4 160 ->
    // 157 in the window, 170 -> 164 sustained, one forced re-binning in 3 000 steps; K = 8 with 0.28 h: 17 of them, 263 us/step).
    // Up to 20 k particles K = 16 stays: K = 24 on the same skin (1.05 h) is 3 % faster on the bench's flow (C2 17.4 -> 16.8
    // us/step, C1 14.2 -> 13.7, no forced re-binning in 12 000 developed steps) but a flow AT the reference's U_max drifts
    // 0.023 h a step -- 23 steps of that are the whole half-skin, and tests/test_gpu_headline_parity.py's state forced 16
    // re-binnings in 50 steps with it (K = 32: 86 in the bench's own flow, 50 us/step); 20 k - 300 k stays at 8 (every other
    // pair measured slower there: profiles/r04_k_skin_c3.txt, _194k.txt).
    // Who re-bins: the device by itself from 2 x 10^6 particles (round 4; 10^6 before the wider skins).  Measured with K = 10 /
    // 0.42 h on the host's schedule against K = 24 / 0.28 h on the device's (window / sustained us/step, profiles/
    // r04_static_vs_dyn_*.txt): 1.25 M particles 338 / 353 against 337-340 / 381-384, 2.5 M 664 / 685 against 636 / 705,
    // 6 M 1 526 / 1 564 against 1 407 / 1 542 -- the self-skipping launches of a device-decided step are 7 % of a step at
    // 1.25 M and 1.7 % at 6 M, the thin skin's shorter lists are worth more the longer the passes.
    const bool dyn_wanted = prm->dynamic_rebin == 1 || (prm->dynamic_rebin == 0 && nf >= 2000000);
    const bool auto_policy = prm->rebuild_every <= 0 && prm->skin_h <= 0.0;
    int K = prm->rebuild_every > 0 ? std::min(prm->rebuild_every, 64) : (nf <= 20000 ? 16 : (nf <= 300000 ? 8 : (dyn_wanted ? 24 : 10)));
    // dual-rate loop: a slot moves particles n_sub times as far.  Same eligibility as ctx_alloc's n_in (the fused E|A launch
    // with the clock in its tail workgroup: a skin, static schedule, <= 2048 workgroups, compact kernels)
    const bool dual_ok = c->lpp >= 16 && !c->is_slab && K > 1 && prm->dynamic_rebin != 1 &&
                         div_up((size_t)nf * c->lpp, kBlock) <= (size_t)tail_clock_limit() && !debug_switches().no_tail_clock && !debug_switches().no_fuse_ea;
    const int n_sub = dual_ok ? dual_rate_substeps(*prm) : 1;
    if (n_sub > 1 && prm->rebuild_every <= 0) K = std::max(2, K / n_sub);
    const double d_step = 0.035 * prm->h * n_sub;
    double skin = K > 1 ? (prm->skin_h > 0.0 ? prm->skin_h * prm->h : 2.0 * std::max((K - 1) * d_step, 0.1 * prm->h)) : 0.0;
    if (auto_policy && nf > 300000 && K > 1) skin = (dyn_wanted ? 0.28 : 0.42) * prm->h;  // (see above: not the skin of this K)
    if (K > 1 && (int)std::floor(prm->DL / (2.0 * prm->h + skin)) < 3) { K = 1; skin = 0.0; }
    c->rebuild_every = K;
    c->skin = skin;
    // Dynamic re-binning pays where a handful of empty launches per step is noise next to the passes (measured: the
    // host-driven forced rebuilds + cool-downs cost 10-25 % of the sustained rate from 0.5 M particles up).
    // dynamic_rebin: 0 = by size, 1 = on, 2 = off.
    c->dyn = skin > 0.0 && dyn_wanted;

    // grid: exact periodic tiling in x (cells >= 2h + skin), rows of 2h + skin in y over fluid + wall extent
    double y_min, y_max;
    y_extent(py, n_total, y_min, y_max);
    const double cs = 2.0 * prm->h + skin;
    Grid g{};
    // Any DL > 0 (the reference's grid is ceil(DL/2h) >= 1 columns plus ghost entries, neighbor.c:253-296): below three
    // columns the sweeps visit each distinct column once (duplicate_column) and the fold picks the nearest image.
    g.ncx = std::max(1, (int)std::floor(prm->DL / cs));
    g.ncy = (int)std::ceil((y_max - y_min + 1e-12) / cs) + 1;
    require((double)g.ncx * (double)g.ncy < 2.0e9, "SPH:Neighbor:param", "cell grid too large.");
    g.ncells = g.ncx * g.ncy;
    g.periodic = 1;
    g.DL = prm->DL;
    g.half_DL = 0.5 * prm->DL;
    g.x0 = 0.0;
    g.y0 = y_min;
    g.inv_csx = (double)g.ncx / prm->DL;
    g.inv_csy = 1.0 / cs;
    g.own_lo = -std::numeric_limits<double>::infinity();
    g.own_hi = std::numeric_limits<double>::infinity();
    c->grid = g;
    c->phys = make_phys(prm);
    choose_kernel_forms(c, prm->lanes_per_particle > 0, (double)nf / g.ncx, nf);

    SPHX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    SPHX_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_clock), sizeof(Clock), hipHostMallocDefault));
    ctx_alloc(c, nf);
    upload_fluid(c, nf, px, py, vel, vel + ntz, drho_dt, mass, nullptr, true);
    std::vector<int> wid((size_t)std::max(nw, 1));
    for (int k = 0; k < nw; ++k) wid[k] = nf + k;
    upload_walls(c, nw, px + nf, py + nf, mass + nf, wall_vel + nf, wall_vel + ntz + nf, wid.data(), true);
    init_clock(c, nf, t0, step0);
    read_clock(c);
    // capture the step graphs now, not inside somebody's timed region (a few ms) -- unless this context can never
    // step (the one-shot contexts behind sphx_neighbor_search)
    if (prm->t_end > t0) (void)get_graph(c, c->cur, c->lay, c->pos, graph_slots(c));
}

// emit the MEX-convention pair list of the current ordering into the ctx-held buffers
void emit_pairs(sphx_ctx *c, bool fill)
{
    const int nf = c->nf;
    hipStream_t s = c->stream;
    DevBuf<int> cnt(nf), off((size_t)nf + 1);
    const FluidSet fs = c->view(c->cur, c->lay);
    const Clock *clk = c->clock.get();
    const dim3 g1(div_up(nf, kBlock)), b1(kBlock);
    hipLaunchKernelGGL(k_pairs<0>, g1, b1, 0, s, clk, c->grid, c->phys, fs, c->walls, cnt.get(), (const int *)nullptr,
                       (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr,
                       (double *)nullptr, (double *)nullptr);
    hipLaunchKernelGGL(k_scan_only, dim3(1), dim3(kScanBlock), 0, s, (const Clock *)nullptr, 0, (const int *)cnt.get(), off.get(), nf);
    int total = 0;
    SPHX_HIP(hipMemcpyAsync(&total, off.get() + nf, sizeof(int), hipMemcpyDeviceToHost, s));
    SPHX_HIP(hipStreamSynchronize(s));
    c->pl_n = (size_t)total;
    if (!fill) return;
    const size_t m = c->pl_n ? c->pl_n : 1;
    c->pl_i.alloc(m); c->pl_j.alloc(m); c->pl_dx.alloc(m); c->pl_dy.alloc(m); c->pl_r.alloc(m); c->pl_W.alloc(m); c->pl_dW.alloc(m);
    hipLaunchKernelGGL(k_pairs<1>, g1, b1, 0, s, clk, c->grid, c->phys, fs, c->walls, (int *)nullptr, (const int *)off.get(),
                       c->pl_i.get(), c->pl_j.get(), c->pl_dx.get(), c->pl_dy.get(), c->pl_r.get(), c->pl_W.get(), c->pl_dW.get());
    SPHX_HIP(hipGetLastError());
    SPHX_HIP(hipStreamSynchronize(s));
    c->pl_valid = true;
}

// time one step slot advances by, before clipping to the target (dual-rate loop: n_in sub-steps, see next_dt)
double host_dt_unclipped(const sphx_ctx *c, double vmax)
{
    const double hh = c->phys.kc.h;
    const double dt_ac = 0.25 * hh / std::max(c->phys.c_f + vmax, 1e-12);
    const double dt_rest = std::min(0.125 * hh * hh / std::max(c->phys.nu, 1e-12),
                                    0.25 * std::sqrt(hh / std::max(std::fabs(c->phys.g), 1e-12)));
    return std::min(c->n_in * dt_ac, dt_rest);
}

void throw_on_status(const sphx_ctx *c)
{
    if (c->h_clock->status == SPHX_ERR_GRID)
        throw Error(SPHX_ERR_GRID, "SPHX:Ctx:grid", "neighbour list / slab buffer capacity exceeded on device");
    if (c->h_clock->status != 0)
        throw Error(c->h_clock->status, "SPHX:Ctx:diverged", "device step loop raised a status (non-finite velocity)");
}

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
    if (ctx->comm) (void)sphx_slab_comm_destroy(ctx);
    if (g_fetch_src == ctx) g_fetch_src = nullptr;
    delete ctx;
}

SPHX_EXPORT int sphx_ctx_advance(sphx_ctx *c, double t_target, int64_t max_steps, sphx_status *status)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    require(!c->is_slab, "SPHX:Ctx:slab", "slab contexts are advanced with sphx_slab_compute/finish");
    read_clock(c);  // steps enqueued with sphx_ctx_enqueue_steps may still be in flight
    for (int guard = 0; guard < 1000000; ++guard) {
        if (c->h_clock->need_rebuild && c->h_clock->status == 0) forced_rebuild(c);
        arm_clock(c, t_target, (long long)max_steps, c->cur, (const double *)nullptr);
        // how many slots this call needs, from the unclipped dt; over-provision to whole graphs when unlimited
        const Clock &k = *c->h_clock;
        const double dt_est = host_dt_unclipped(c, k.vmax);
        double want = std::ceil(std::max(0.0, std::min(t_target, k.t_end) - k.t) / std::max(dt_est, 1e-12)) + 1.0;
        // Slots behind a stop (target reached, stale grid) are empty launches, ~10 us each: look at the clock
        // often while the device keeps stopping for forced rebuilds, rarely (4096 slots) once it runs through.
        want = std::min(want, (double)c->chunk_slots);
        int64_t slots = (int64_t)want;
        const int per_graph = graph_slots(c);
        bool exact = false;
        if (max_steps > 0 && slots >= max_steps) { slots = max_steps; exact = true; }  // exact: no no-op slots
        else if (!c->profiling && slots > per_graph) slots = ((slots + per_graph - 1) / per_graph) * per_graph;
        if (slots < 1) slots = 1;
        const int64_t step_before = c->h_clock->step;
        enqueue_slots(c, slots, exact);
        read_clock(c);
        const int64_t executed = c->h_clock->step - step_before;
        // ... and in between by how often it has been stopping: behind a stop the rest of the chunk drains as empty launches
        // (half a 4096-slot chunk: ~35 ms at 0.5 M particles, measured), a look at the clock costs ~40 us -- chunks of 1/16 of
        // the steps since the last forced rebuild keep both near 0.5 us per step.
        const int64_t cap = c->n_forced_rebuilds == 0
                                ? 4096
                                : std::clamp<int64_t>((c->h_clock->step - c->last_forced_step) / 16, 64, 4096);
        c->chunk_slots = c->h_clock->need_rebuild ? 64 : std::min<int64_t>(cap, 2 * c->chunk_slots);
        if (max_steps > 0) {
            max_steps -= executed;
            if (max_steps <= 0) break;
        }
        if (c->h_clock->status != 0) break;
        if (!(c->h_clock->t < t_target - 1e-12)) break;
        if (!(c->h_clock->t < c->h_clock->t_end - 1e-12)) break;
    }
    c->pending_target = c->h_clock->step;
    fill_status(c, status);
    throw_on_status(c);
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_enqueue_steps(sphx_ctx *c, int64_t n_steps)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    require(!c->is_slab, "SPHX:Ctx:slab", "slab contexts are advanced with sphx_slab_compute/finish");
    require(n_steps > 0, "SPHX:Ctx:steps", "n_steps must be positive");
    // no host sync here: if an earlier batch stopped early (end time, status, stale grid) the loop condition is
    // still false when k_prepare re-evaluates it, so every slot of this batch is a no-op; sphx_ctx_sync sorts
    // out the grid and takes the steps that are still owed
    arm_clock(c, c->prm.t_end, (long long)n_steps, c->cur, (const double *)nullptr);
    enqueue_slots(c, n_steps, true);
    c->pending_target = std::max<int64_t>(c->pending_target, c->h_clock->step) + n_steps;
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_prepare_steps(sphx_ctx *c, int64_t n_steps)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    require(!c->is_slab, "SPHX:Ctx:slab", "slab contexts are advanced with sphx_slab_run");
    require(n_steps > 0, "SPHX:Ctx:steps", "n_steps must be positive");
    read_clock(c);  // the phase the next batch starts from
    if (c->h_clock->need_rebuild && c->h_clock->status == 0) forced_rebuild(c);
    hipLaunchKernelGGL(k_disarm, dim3(1), dim3(1), 0, c->stream, c->clock.get());  // (every batch re-arms with k_prepare)
    enqueue_slots(c, n_steps, true, true);
    SPHX_HIP(hipStreamSynchronize(c->stream));
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_graph_stats(sphx_ctx *c, int64_t *slots_replayed, int64_t *slots_eager, int64_t *graphs_captured)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    if (slots_replayed) *slots_replayed = c->slots_replayed;
    if (slots_eager) *slots_eager = c->slots_eager;
    if (graphs_captured) *graphs_captured = c->graphs_captured;
    return SPHX_OK;
    SPHX_CATCH
}

// Wait for what is enqueued and take the steps a batch of sphx_ctx_enqueue_steps still owes: a batch stops early when the
// drift bound is hit (static schedule) -- re-bin, then go on, in chunks (slots behind another stop would be empty launches).
// Leaves the clock read.  Whoever hands out results calls this first: a batch that stopped before its armed last step has
// not written the output-only fields of that step (FluidTmp::lazy_out), so force / rho / p would be those of an older batch.
static void settle_owed(sphx_ctx *c)
{
    read_clock(c);
    for (int guard = 0; guard < 1000000; ++guard) {
        const Clock &k = *c->h_clock;
        const int64_t owed = c->pending_target - (int64_t)k.step;
        if (!(k.status == 0 && owed > 0 && k.t < k.t_end - 1e-12)) break;
        if (k.need_rebuild) {
            forced_rebuild(c);
            c->chunk_slots = 64;
        }
        const int64_t n = std::min(owed, c->chunk_slots);
        arm_clock(c, c->prm.t_end, (long long)n, c->cur, (const double *)nullptr);
        enqueue_slots(c, n, false);
        read_clock(c);
        if (!c->h_clock->need_rebuild) c->chunk_slots = std::min<int64_t>(4096, 2 * c->chunk_slots);
    }
    c->pending_target = c->h_clock->step;
}

SPHX_EXPORT int sphx_ctx_sync(sphx_ctx *c, sphx_status *status)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    require(!c->is_slab, "SPHX:Ctx:slab", "use sphx_slab_sync on a slab context");
    settle_owed(c);
    fill_status(c, status);
    throw_on_status(c);
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_download(sphx_ctx *c, double *pos, double *vel, double *rho, double *p, double *drho_dt,
                                  double *force, double *force_prior, double *Vol, double *B)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    require(!c->is_slab, "SPHX:Ctx:slab", "use sphx_slab_snapshot on a slab context");
    settle_owed(c);  // (state and step outputs of the SAME step: the last one of everything enqueued, see settle_owed)
    const bool need_outputs = rho || p || force || force_prior || Vol || B;
    if (need_outputs && !c->have_step_outputs)
        throw Error(SPHX_ERR_STATE, "SPHX:Ctx:download", "rho/p/force/Vol/B exist only after at least one step");
    const int nf = c->nf, nw = c->nw, nt = c->nt;
    hipStream_t s = c->stream;
    const FluidSet fs = c->view(c->cur, c->lay);
    // ordering the step outputs are stored in: the layout the step ran in (static schedule), or -- dynamic contexts,
    // which re-bin in place -- the current ids reached through src_of when the last step ended with a re-binning
    const int *id_old = c->dyn ? fs.id : c->fid_[c->out_lay].get();
    const int *out_map = (c->dyn && c->h_clock->fresh) ? (const int *)c->src_of.get() : nullptr;
    DevBuf<double> stage((size_t)4 * nt);
    const dim3 gf(div_up(nf, kBlock)), gw(div_up(std::max(nw, 1), kBlock)), b(kBlock);
    auto col = [&](int cidx) { return stage.get() + (size_t)cidx * nt; };
    // component `comp` of a field stored as records of `stride` doubles
    auto unsort_f = [&](const int *id, const void *src, int stride, int comp, int cidx) {  // current state
        hipLaunchKernelGGL(k_unsort, gf, b, 0, s, nf, id, (const double *)src + comp, stride, col(cidx), (const int *)nullptr);
    };
    auto unsort_o = [&](const void *src, int stride, int comp, int cidx) {  // outputs of the last step
        hipLaunchKernelGGL(k_unsort, gf, b, 0, s, nf, id_old, (const double *)src + comp, stride, col(cidx), out_map);
    };
    auto unsort_w = [&](const void *src, int stride, int comp, int cidx) {
        if (nw > 0)
            hipLaunchKernelGGL(k_unsort, gw, b, 0, s, nw, (const int *)c->wid.get(), (const double *)src + comp, stride, col(cidx),
                               (const int *)nullptr);
    };
    auto fill_w = [&](int cidx, double v) {
        if (nw > 0) hipLaunchKernelGGL(k_fill, gw, b, 0, s, nw, col(cidx) + nf, v);
    };
    auto out = [&](double *host, int ncol) {
        SPHX_HIP(hipMemcpyAsync(host, stage.get(), (size_t)ncol * nt * sizeof(double), hipMemcpyDeviceToHost, s));
        SPHX_HIP(hipStreamSynchronize(s));
    };
    if (pos) {
        unsort_f(fs.id, fs.pos, 2, 0, 0); unsort_f(fs.id, fs.pos, 2, 1, 1);
        unsort_w(c->wpos.get(), 2, 0, 0); unsort_w(c->wpos.get(), 2, 1, 1);
        out(pos, 2);
    }
    if (vel) { unsort_f(fs.id, fs.vel, 2, 0, 0); unsort_f(fs.id, fs.vel, 2, 1, 1); fill_w(0, 0.0); fill_w(1, 0.0); out(vel, 2); }
    if (drho_dt) { unsort_f(fs.id, fs.drho, 1, 0, 0); fill_w(0, 0.0); out(drho_dt, 1); }
    if (rho) { unsort_o(c->rho_out.get(), 1, 0, 0); fill_w(0, c->prm.rho0); out(rho, 1); }
    if (p) { unsort_o(c->p_out.get(), 1, 0, 0); fill_w(0, 0.0); out(p, 1); }
    if (force) { unsort_o(c->ff.get(), 2, 0, 0); unsort_o(c->ff.get(), 2, 1, 1); fill_w(0, 0.0); fill_w(1, 0.0); out(force, 2); }
    if (force_prior) { unsort_o(c->ffp.get(), 2, 0, 0); unsort_o(c->ffp.get(), 2, 1, 1); fill_w(0, 0.0); fill_w(1, 0.0); out(force_prior, 2); }
    if (Vol) { unsort_o(c->tmp_par[c->fuse_ea ? c->out_par : 0].a, 4, 0, 0); unsort_w(c->wa.get(), 4, 0, 0); out(Vol, 1); }
    if (B) {
        for (int k = 0; k < 4; ++k) unsort_o(c->fB.get(), 4, k, k);
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
    require(!c->is_slab, "SPHX:Ctx:slab", "monitors are not available on a slab context");
    settle_owed(c);
    if (c->h_clock->need_rebuild && c->h_clock->status == 0) forced_rebuild(c);  // the pair sweeps need valid bins
    hipStream_t s = c->stream;
    const FluidSet fs = c->view(c->cur, c->lay);
    if (tau_bottom || tau_top) {
        if (!c->have_step_outputs)
            throw Error(SPHX_ERR_STATE, "SPHX:Ctx:monitor", "wall shear needs Vol/B of a completed step");
        const int nblk = c->n_blocks_flat;
        hipLaunchKernelGGL(k_wall_shear, dim3(nblk), dim3(kBlock), 0, s, (const Clock *)c->clock.get(), c->grid, c->phys, fs,
                           c->tmp_par[c->fuse_ea ? c->out_par : 0], c->walls,
                           (c->dyn ? c->h_clock->fresh != 0 : c->out_lay != c->lay) ? 1 : 0, c->tau_part.get());
        hipLaunchKernelGGL(k_tau_final, dim3(1), dim3(kScanBlock), 0, s, nblk, (const double *)c->tau_part.get(),
                           c->phys.DL, c->tau_out.get());
        double h[2];
        SPHX_HIP(hipMemcpyAsync(h, c->tau_out.get(), sizeof(h), hipMemcpyDeviceToHost, s));
        SPHX_HIP(hipStreamSynchronize(s));
        if (tau_bottom) *tau_bottom = h[0];
        if (tau_top) *tau_top = h[1];
    }
    if (n_pairs) {
        emit_pairs(c, false);
        *n_pairs = (double)c->pl_n;
    }
    SPHX_HIP(hipGetLastError());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_neighbor_list(sphx_ctx *c, size_t *n_pairs)
{
    SPHX_TRY
    require(c != nullptr && n_pairs != nullptr, "SPHX:Ctx:null", "ctx / n_pairs must not be NULL");
    require(!c->is_slab, "SPHX:Ctx:slab", "pair lists are not available on a slab context");
    read_clock(c);
    if (c->h_clock->need_rebuild && c->h_clock->status == 0) forced_rebuild(c);
    emit_pairs(c, true);
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
        prm.rebuild_every = 1;
        const size_t nt = (size_t)n_total;
        std::vector<double> zeros2(2 * nt, 0.0), ones(nt, 1.0);
        // The driver calls this once per step with the same sizes (SPH_Poiseuille.m:428): keep the search context
        // (stream, pinned clock, device arrays, wall grid shape) and only reload the positions.  Creating and
        // destroying it every call cost 4.1 ms at 5 760 particles, the search itself 0.4 ms.
        sphx_ctx *old = g_search_ctx;
        bool reuse = false;
        if (old && old->nf == n_fluid && old->nt == n_total && old->prm.h == h && old->prm.DL == DL) {
            double y_min, y_max;
            y_extent(pos + nt, n_total, y_min, y_max);
            const double cs = 2.0 * h;
            reuse = old->grid.y0 == y_min && old->grid.ncy == (int)std::ceil((y_max - y_min + 1e-12) / cs) + 1;
        }
        if (reuse) {
            c = old;
            const int nw = n_total - n_fluid;
            upload_fluid(c, n_fluid, pos, pos + nt, zeros2.data(), zeros2.data(), zeros2.data(), ones.data(), nullptr, true);
            std::vector<int> wid((size_t)std::max(nw, 1));
            for (int k = 0; k < nw; ++k) wid[k] = n_fluid + k;
            upload_walls(c, nw, pos + n_fluid, pos + nt + n_fluid, ones.data(), zeros2.data(), zeros2.data(), wid.data(), true);
        } else {
            c = new sphx_ctx();
            ctx_setup(c, &prm, n_fluid, n_total, pos, zeros2.data(), zeros2.data(), ones.data(), zeros2.data(), 0.0, 0);
        }
        emit_pairs(c, true);
        *n_pairs = c->pl_n;
        if (old && old != c) delete old;
        g_search_ctx = c;
        g_fetch_src = c;
        return SPHX_OK;
    } catch (const Error &e) {
        if (c == g_search_ctx) g_search_ctx = nullptr;
        if (g_fetch_src == c) g_fetch_src = nullptr;
        delete c;
        return report(e);
    } catch (const std::exception &e) {
        if (c == g_search_ctx) g_search_ctx = nullptr;
        if (g_fetch_src == c) g_fetch_src = nullptr;
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
    g_fetch_src = nullptr;  // the search context itself stays for the next search of the same shape
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
    c->timer.timer_log = debug_switches().log;
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

SPHX_EXPORT int sphx_ctx_grid_policy(sphx_ctx *c, int *rebuild_every, double *skin, int64_t *forced_rebuilds,
                                     double *drift)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    if (rebuild_every) *rebuild_every = c->rebuild_every;
    if (skin) *skin = c->skin;
    if (forced_rebuilds) *forced_rebuilds = c->dyn ? (int64_t)c->h_clock->n_drift_rebuilds : c->n_forced_rebuilds;
    if (drift) *drift = c->h_clock->drift;
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

SPHX_EXPORT int sphx_ctx_schedule(sphx_ctx *c, int *fuse_ea, int *tail_clock, int *dynamic, int64_t *rebins)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    if (fuse_ea) *fuse_ea = c->fuse_ea ? 1 : 0;
    if (tail_clock) *tail_clock = c->tail_clock ? 1 : 0;
    if (dynamic) *dynamic = c->dyn ? 1 : 0;
    if (rebins) *rebins = c->dyn ? (int64_t)c->h_clock->n_rebins : c->n_rebins;
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_kernel_forms(sphx_ctx *c, int *walk_kernels, int *lds_tiles, int *tiles_abe, int *coded_lists)
{
    SPHX_TRY
    require(c != nullptr, "SPHX:Ctx:null", "ctx must not be NULL");
    if (walk_kernels) *walk_kernels = c->walk_kernels ? 1 : 0;
    if (lds_tiles) *lds_tiles = c->lds_tiles ? 1 : 0;
    if (tiles_abe) *tiles_abe = c->lds_tiles_be ? 1 : 0;
    if (coded_lists) *coded_lists = c->coded_lists ? 1 : 0;
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_ctx_substeps(sphx_ctx *c, int *n_inner)
{
    SPHX_TRY
    require(c != nullptr && n_inner != nullptr, "SPHX:Ctx:null", "ctx / n_inner must not be NULL");
    *n_inner = c->n_in;
    return SPHX_OK;
    SPHX_CATCH
}

// =================================================================================================
// x-slab contexts (multi-GPU): one context per rank holds the particles of its column range plus
// `halo_cols` columns of copies on either side, in an OPEN grid window.  Per step:
//   sphx_slab_compute : k_density..k_continuity on everything held, local max|v| of owned particles,
//                       pack keep / send-left / send-right
//   (caller: exchange the two messages with the ring neighbours, all-reduce max|v|)
//   sphx_slab_finish  : unpack, clock update with the global max|v|, cell rebuild.
// Validity: with halo H columns (each >= 2h wide) pass A is exact on columns >= own-(H-1), pass B on
// >= own-(H-2), the force pass on >= own-(H-3) and the continuity pass on the owned columns for H = 4;
// whatever is computed on halo copies is discarded (their owner sends fresh state).
// =================================================================================================
namespace {

struct HostParticles {
    std::vector<double> x, y, vx, vy, drho, mass;
    std::vector<int> id;
    void push(double X, double Y, double VX, double VY, double D, double M, int I)
    {
        x.push_back(X); y.push_back(Y); vx.push_back(VX); vy.push_back(VY); drho.push_back(D); mass.push_back(M); id.push_back(I);
    }
};

void slab_setup(sphx_ctx *c, const sphx_params *prm, int n_fluid, int n_total, const double *pos, const double *vel,
                const double *drho_dt, const double *mass, const double *wall_vel, double t0, int64_t step0, int rank,
                int n_ranks, int halo_cols, void *stream)
{
    common_checks(prm, n_fluid, n_total);
    require(n_ranks >= 2 && rank >= 0 && rank < n_ranks, "SPHX:Slab:rank", "slab contexts need n_ranks >= 2 and 0 <= rank < n_ranks");
    require(halo_cols >= 4, "SPHX:Slab:halo", "halo_cols must be >= 4 (four dependent neighbour passes per step)");
    ensure_device();
    c->prm = *prm;
    c->is_slab = true;
    c->rank = rank; c->n_ranks = n_ranks; c->halo_cols = halo_cols;
    c->nf = n_fluid; c->nt = n_total; c->nw = n_total - n_fluid;
    const size_t ntz = (size_t)n_total;
    const double *px = pos, *py = pos + ntz;
    double y_min, y_max;
    y_extent(py, n_total, y_min, y_max);
    // Re-binning interval of the slab: 1 = every step (the protocol of sphx_slab_compute / _finish, a caller-driven
    // transport); K > 1 (0 = auto: 5) = every K-th step with a cell skin, frozen layouts and fixed exchange lists in between
    // (the native loops sphx_slab_run / sphx_slab_group_run only).
    // (round 4: a slab decides on the device like a dynamic context -- the skin of K = 5, scheduled re-binnings every 24th step,
    //  see ctx_setup)
    const bool auto_policy = prm->rebuild_every <= 0 && prm->skin_h <= 0.0;
    int K = prm->rebuild_every > 0 ? std::min(prm->rebuild_every, 64) : 24;
    const double d_step = 0.035 * prm->h;
    double skin = K > 1 ? (prm->skin_h > 0.0 ? prm->skin_h * prm->h : 2.0 * std::max((K - 1) * d_step, 0.1 * prm->h)) : 0.0;
    if (auto_policy) skin = 0.28 * prm->h;
    c->rebuild_every = K;
    c->skin = skin;
    c->dyn = K > 1;  // the device decides when to re-bin (from all-reduced maxima), the layout is rebuilt in place
    const double cs = 2.0 * prm->h + skin, DL = prm->DL;
    const int ncx_g = (int)std::floor(DL / cs);
    require(ncx_g >= 3, "SPH:Neighbor:param", "channel too short for three cell columns.");
    const double csx = DL / ncx_g;
    const int H = halo_cols;
    auto col_lo = [&](int r) { return (int)(((long long)r * ncx_g) / n_ranks); };
    c->col0 = col_lo(rank);
    c->col1 = col_lo(rank + 1);
    for (int r = 0; r < n_ranks; ++r)
        require(col_lo(r + 1) - col_lo(r) >= H + 1, "SPHX:Slab:width",
                "every slab must own at least halo_cols+1 cell columns (use fewer ranks or a longer channel)");
    require((c->col1 - c->col0) + 2 * H < ncx_g, "SPHX:Slab:width", "slab window would cover the whole period");

    Grid g{};
    g.ncx = (c->col1 - c->col0) + 2 * H;
    g.ncy = (int)std::ceil((y_max - y_min + 1e-12) / cs) + 1;
    g.ncells = g.ncx * g.ncy;
    g.periodic = 0;
    g.DL = DL;
    g.half_DL = std::numeric_limits<double>::infinity();
    g.x0 = (double)(c->col0 - H) * csx;
    g.y0 = y_min;
    g.inv_csx = (double)ncx_g / DL;
    g.inv_csy = 1.0 / cs;
    g.own_lo = (double)c->col0 * csx;
    g.own_hi = (rank == n_ranks - 1) ? DL : (double)c->col1 * csx;
    g.own_by_cell = K > 1 ? 1 : 0;
    g.own_c0 = H;
    g.own_c1 = H + (c->col1 - c->col0);
    c->grid = g;
    c->phys = make_phys(prm);
    const double win_lo = g.x0, win_hi = g.x0 + (double)g.ncx * csx;

    // select the particles (and periodic images) inside this rank's window
    HostParticles hf, hw;
    std::vector<double> wvx, wvy;
    for (int i = 0; i < n_total; ++i) {
        const double xw = px[i] - std::floor(px[i] / DL) * DL;
        for (int im = -1; im <= 1; ++im) {
            const double xs = xw + im * DL;
            if (!(xs >= win_lo && xs < win_hi)) continue;
            if (i < n_fluid) hf.push(xs, py[i], vel[i], vel[ntz + i], drho_dt[i], mass[i], i);
            else { hw.push(xs, py[i], 0, 0, 0, mass[i], i); wvx.push_back(wall_vel[i]); wvy.push_back(wall_vel[ntz + i]); }
        }
    }
    const int n_local = (int)hf.x.size(), nw_local = (int)hw.x.size();
    // capacities: a message carries the H boundary columns plus what crossed in one step -- (H+1) columns at 1.3x the mean
    // column load (the whole buffer travels every step: 1 + 7 cap doubles per neighbour; overflow raises SPHX_ERR_GRID)
    const double per_col = (double)n_fluid / ncx_g;
    c->msg_cap = (int)(1.3 * (H + 1) * per_col) + 1024;
    // (every kernel is launched for `cap` particles: slack costs time -- 15 % headroom over the initial population of this
    // weakly compressible flow; an overflow raises SPHX_ERR_GRID on the device, never a silent loss)
    const int cap = (int)(1.15 * std::max<double>(n_local, per_col * g.ncx)) + 2 * c->msg_cap + 1024;
    require(n_local <= cap, "SPHX:Slab:capacity", "initial slab population exceeds capacity");

    c->lpp = prm->lanes_per_particle > 0 ? prm->lanes_per_particle : pick_lpp(n_local);
    check_lpp(c->lpp);
    choose_kernel_forms(c, prm->lanes_per_particle > 0, per_col, n_local);
    c->spg = 2;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else SPHX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    SPHX_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_clock), sizeof(Clock), hipHostMallocDefault));
    ctx_alloc(c, cap);
    upload_fluid(c, n_local, hf.x.data(), hf.y.data(), hf.vx.data(), hf.vy.data(), hf.drho.data(), hf.mass.data(),
                 hf.id.data(), false);
    upload_walls(c, nw_local, hw.x.data(), hw.y.data(), hw.mass.data(), wvx.data(), wvy.data(), hw.id.data(), false);

    c->kpos.alloc(cap); c->kvel.alloc(cap); c->kdrho.alloc(cap); c->kmass.alloc(cap);
    c->kid.alloc(cap); c->counters.alloc(3); c->n_new.alloc(1); c->ticket.alloc(1);
    c->counters.zero(c->stream); c->ticket.zero(c->stream); c->n_new.zero(c->stream);
    SlabPack p{};
    p.counters = c->counters.get();
    p.kpos = c->kpos.get(); p.kvel = c->kvel.get(); p.kdrho = c->kdrho.get();
    p.kmass = c->kmass.get(); p.kid = c->kid.get();
    p.cellid = c->cellid.get(); p.count = c->count.get();
    p.halo_w = (double)H * csx;
    p.shift_l = (rank == 0) ? DL : 0.0;             // my left neighbour is the last slab: it sees me at x + DL
    p.shift_r = (rank == n_ranks - 1) ? -DL : 0.0;  // my right neighbour is the first slab
    p.win_lo = win_lo;
    p.win_hi = win_hi;
    p.msg_cap = c->msg_cap;
    p.keep_cap = cap;
    c->pack = p;
    if (K > 1) {
        for (int k = 0; k < 2; ++k) {
            c->send_idx_[k].alloc(c->msg_cap); c->recv_slot_[k].alloc(c->msg_cap);
            c->ids_s_[k].alloc((size_t)c->msg_cap + 1); c->ids_r_[k].alloc((size_t)c->msg_cap + 1);
            c->ids_s_[k].zero(c->stream); c->ids_r_[k].zero(c->stream);
        }
        c->send_cnt.alloc(2); c->recv_cnt.alloc(2); c->slot_of_id.alloc((size_t)n_fluid);
        c->send_cnt.zero(c->stream); c->recv_cnt.zero(c->stream);
        SPHX_HIP(hipMemsetAsync(c->slot_of_id.get(), 0xFF, (size_t)n_fluid * sizeof(int), c->stream));
        c->lists = SlabLists{{c->send_idx_[0].get(), c->send_idx_[1].get()}, c->send_cnt.get(),
                             {c->recv_slot_[0].get(), c->recv_slot_[1].get()}, c->recv_cnt.get(), c->slot_of_id.get(),
                             {c->ids_s_[0].get(), c->ids_s_[1].get()}};
        if (n_local > 0)
            hipLaunchKernelGGL(k_slot_of_id, dim3(div_up(n_local, kBlock)), dim3(kBlock), 0, c->stream, n_local,
                               (const int *)c->fid_[0].get(), c->slot_of_id.get());
        SPHX_HIP(hipGetLastError());
    }
    init_clock(c, n_local, t0, step0);
    c->slab_step0 = step0;
    read_clock(c);
}

}  // namespace

namespace {

// run one half of a slab step (whole steps are replayed as graphs by sphx_slab_graph_prepare; graphs of half a step --
// the exchange sits between the halves -- cost more host time than the ten plain launches they held)
template <typename Body>
void slab_half(sphx_ctx *, int, int, const void *const[3], Body &&body) { body(); }

}  // namespace

SPHX_EXPORT int sphx_slab_create(sphx_ctx **out, const sphx_params *prm, int n_fluid, int n_total, const double *pos,
                                 const double *vel, const double *drho_dt, const double *mass, const double *wall_vel,
                                 double t0, int64_t step0, int rank, int n_ranks, int halo_cols, void *hip_stream)
{
    sphx_ctx *c = nullptr;
    try {
        require(out != nullptr, "SPHX:Ctx:out", "ctx output pointer must not be NULL");
        c = new sphx_ctx();
        slab_setup(c, prm, n_fluid, n_total, pos, vel, drho_dt, mass, wall_vel, t0, step0, rank, n_ranks, halo_cols, hip_stream);
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

SPHX_EXPORT int sphx_slab_layout(sphx_ctx *c, int64_t *msg_doubles, int *col0, int *col1, int *n_local, int *capacity)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab, "SPHX:Slab:ctx", "not a slab context");
    if (msg_doubles) *msg_doubles = 1 + 7 * (int64_t)c->msg_cap;
    if (col0) *col0 = c->col0;
    if (col1) *col1 = c->col1;
    if (n_local) *n_local = c->h_clock->n;
    if (capacity) *capacity = c->cap;
    return SPHX_OK;
    SPHX_CATCH
}

// local max |v| of owned particles at the current state -> vmax_dev[0] (device pointer); async.
SPHX_EXPORT int sphx_slab_local_vmax(sphx_ctx *c, double *vmax_dev)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab && vmax_dev != nullptr, "SPHX:Slab:ctx", "not a slab context");
    require(c->rebuild_every == 1, "SPHX:Slab:protocol", "local_vmax / prepare / compute / finish drive a slab created with rebuild_every = 1");
    const FluidSet fs = c->view(c->cur, c->cur);
    hipLaunchKernelGGL(k_vmax_init, dim3(1), dim3(kScanBlock), 0, c->stream, c->clock.get(), c->grid, (const double2 *)fs.pos,
                       (const double2 *)fs.vel, vmax_dev);
    SPHX_HIP(hipGetLastError());
    return SPHX_OK;
    SPHX_CATCH
}

// arm the clock with the all-reduced max |v| (device pointer); async.
SPHX_EXPORT int sphx_slab_prepare(sphx_ctx *c, double t_target, int64_t max_steps, const double *vmax_global_dev)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab, "SPHX:Slab:ctx", "not a slab context");
    require(c->rebuild_every == 1, "SPHX:Slab:protocol", "local_vmax / prepare / compute / finish drive a slab created with rebuild_every = 1");
    arm_clock(c, t_target, (long long)max_steps, c->cur, vmax_global_dev);
    SPHX_HIP(hipGetLastError());
    return SPHX_OK;
    SPHX_CATCH
}

namespace {

// first half of a slab step on the context's stream: the four neighbour passes, pack, seal + local max |v|
void slab_compute_impl(sphx_ctx *c, double *send_left_dev, double *send_right_dev, double *vmax_local_dev)
{
    const int q = c->cur;
    const Clock *clk = c->clock.get();
    auto body = [&]() {
        launch_physics_any(c, q, c->view(q, q), c->tmp, 0);
        SlabPack p = c->pack;
        p.send_l = send_left_dev;
        p.send_r = send_right_dev;
        launch(c, "k_slab_pack", k_slab_pack, dim3(c->n_blocks_flat), dim3(kBlock), clk, q, c->grid, c->view(q, q), c->tmp, p);
        launch(c, "k_slab_seal_vmax", k_slab_seal_vmax, dim3(1), dim3(kScanBlock), clk, q, p, c->n_vpart,
               (const double *)c->vpart.get(), vmax_local_dev);
    };
    const void *key[3] = {send_left_dev, send_right_dev, vmax_local_dev};
    slab_half(c, 0, q, key, body);
    SPHX_HIP(hipGetLastError());
}

// second half: unpack the received messages, clock update with the global max |v|, cell rebuild
void slab_finish_impl(sphx_ctx *c, const double *recv_left_dev, const double *recv_right_dev, const double *vmax_global_dev)
{
    const int q = c->cur;
    Clock *clk = c->clock.get();
    const dim3 bp(kBlock);
    auto body = [&]() {
        // kept particles were binned by the pack kernel, the received ones are binned here
        launch(c, "k_slab_unpack", k_slab_unpack, dim3(div_up((size_t)2 * c->msg_cap, kBlock)), bp, (const Clock *)clk, q,
               c->grid, c->pack, recv_left_dev, recv_right_dev, c->n_new.get(), c->flags.get());
        // clock: t += dt, new particle count, next dt from the global max |v| -- and the scan of the cell histogram
        // and the reset of the pack counters.  It arms run[1-q]; the remaining kernels of this slot still test
        // run[q], and from here on clk->n is the new particle count.
        const FluidSet d = c->view(1 - q, 1 - q);
        if (!c->big_scan) {
            launch(c, "k_clock_scan", k_clock_scan, dim3(1), dim3(kScanBlock), clk, q, c->phys, 0, (const double *)nullptr,
                   vmax_global_dev, (const int *)c->flags.get(), (const int *)c->count.get(), d.start, c->grid.ncells,
                   (const int *)c->n_new.get(), (const double *)nullptr, 1, 0.0, c->counters.get(), (unsigned long long *)nullptr, 0);
        } else {
            int *tile_sum = c->tile.get(), *tile_off = c->tile.get() + c->n_tiles + 1;
            launch(c, "k_scan_tiles", k_scan_tiles, dim3(c->n_tiles), dim3(kScanBlock), (const Clock *)clk, q,
                   (const int *)c->count.get(), d.start, tile_sum, c->grid.ncells);
            launch(c, "k_clock_scan", k_clock_scan, dim3(1), dim3(kScanBlock), clk, q, c->phys, 0, (const double *)nullptr,
                   vmax_global_dev, (const int *)c->flags.get(), (const int *)tile_sum, tile_off, c->n_tiles,
                   (const int *)c->n_new.get(), (const double *)nullptr, 1, 0.0, c->counters.get(), (unsigned long long *)nullptr, 0);
            launch(c, "k_scan_add", k_scan_add, dim3(c->n_tiles), dim3(kScanBlock), (const Clock *)clk, q, d.start,
                   (const int *)tile_off, c->grid.ncells, c->n_tiles);
        }
        launch_scatter_reorder(c, clk, q,
                               reorder_args(c->kpos.get(), c->kvel.get(), c->kdrho.get(), c->kmass.get(), c->kid.get(), d, nullptr), d);
    };
    const void *key[3] = {recv_left_dev, recv_right_dev, vmax_global_dev};
    slab_half(c, 1, q, key, body);
    SPHX_HIP(hipGetLastError());
    c->cur ^= 1;
    c->slab_steps_enqueued += 1;
}

}  // namespace

SPHX_EXPORT int sphx_slab_compute(sphx_ctx *c, double *send_left_dev, double *send_right_dev, double *vmax_local_dev)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab, "SPHX:Slab:ctx", "not a slab context");
    require(send_left_dev && send_right_dev && vmax_local_dev, "SPHX:Slab:buffers", "message / vmax buffers must not be NULL");
    require(c->rebuild_every == 1, "SPHX:Slab:protocol", "compute/finish drive a slab created with rebuild_every = 1");
    slab_compute_impl(c, send_left_dev, send_right_dev, vmax_local_dev);
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_slab_finish(sphx_ctx *c, const double *recv_left_dev, const double *recv_right_dev,
                                 const double *vmax_global_dev)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab, "SPHX:Slab:ctx", "not a slab context");
    require(recv_left_dev && recv_right_dev && vmax_global_dev, "SPHX:Slab:buffers", "message / vmax buffers must not be NULL");
    require(c->rebuild_every == 1, "SPHX:Slab:protocol", "compute/finish drive a slab created with rebuild_every = 1");
    slab_finish_impl(c, recv_left_dev, recv_right_dev, vmax_global_dev);
    return SPHX_OK;
    SPHX_CATCH
}

// =================================================================================================
// Native step loop of the slabs: everything a step needs is enqueued from here, no host language in the loop.
//   sphx_slab_run        one context per process (one process per GPU): messages and the max all-reduce go through
//                        RCCL on the context's stream -- two sends + two receives in one group per step (ring
//                        neighbours: direct xGMI hops) and one 8-byte all-reduce;
//   sphx_slab_group_run  all slabs of the ring live in one process on one device (tests, rehearsals on a one-GPU box):
//                        the same loop with device-to-device copies as the transport and events for the ordering.
// librccl is loaded when the first communicator is made (dlopen: PyTorch carries its own copy of the library and the two
// must not be linked into one image twice), so a box without RCCL can still run everything single-GPU.
// =================================================================================================
namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;

    static Rccl &get()
    {
        static Rccl r;
        static bool ready = false;
        if (ready) return r;
        // resolve into a local table and publish it only when every entry point is there: a failed attempt leaves nothing
        // half-initialised behind (an older librccl lacking a symbol must fail every call, not only the first)
        Rccl t;
        std::string why;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            t.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (t.lib) break;
            if (const char *e = dlerror()) why = e;
        }
        if (!t.lib) throw Error(SPHX_ERR_DEVICE, "SPHX:Slab:rccl", "cannot load librccl: " + why);
        auto sym = [&](const char *n) {
            void *p = dlsym(t.lib, n);
            if (!p) {
                (void)dlclose(t.lib);
                throw Error(SPHX_ERR_DEVICE, "SPHX:Slab:rccl", std::string("librccl lacks ") + n);
            }
            return p;
        };
        t.GetUniqueId = (decltype(t.GetUniqueId))sym("ncclGetUniqueId");
        t.CommInitRank = (decltype(t.CommInitRank))sym("ncclCommInitRank");
        t.CommDestroy = (decltype(t.CommDestroy))sym("ncclCommDestroy");
        t.GroupStart = (decltype(t.GroupStart))sym("ncclGroupStart");
        t.GroupEnd = (decltype(t.GroupEnd))sym("ncclGroupEnd");
        t.Send = (decltype(t.Send))sym("ncclSend");
        t.Recv = (decltype(t.Recv))sym("ncclRecv");
        t.AllReduce = (decltype(t.AllReduce))sym("ncclAllReduce");
        t.GetErrorString = (decltype(t.GetErrorString))sym("ncclGetErrorString");
        r = t;
        ready = true;
        return r;
    }
    void check(ncclResult_t e, const char *what)
    {
        if (e != ncclSuccess) throw Error(SPHX_ERR_DEVICE, "SPHX:Slab:rccl", std::string(what) + ": " + GetErrorString(e));
    }
};

void slab_native_buffers(sphx_ctx *c)
{
    if (c->msg_sl.get()) return;
    const size_t n = 1 + 7 * (size_t)c->msg_cap;
    c->msg_sl.alloc(n); c->msg_sr.alloc(n); c->msg_rl.alloc(n); c->msg_rr.alloc(n);
    c->vmax_l.alloc(2); c->vmax_g.alloc(2);
    for (DevBuf<double> *b : {&c->msg_sl, &c->msg_sr, &c->msg_rl, &c->msg_rr, &c->vmax_l, &c->vmax_g}) b->zero(c->stream);
    // The second stream of a skinned slab (k_slab_maxima beside pass E, the exchange beside the interior of pass A) -- where there
    // is something to hide the latencies behind: the two-stream step has five event hand-overs, one reduction kernel and one
    // launch of pass A more than the chain, ~15 us on the critical path of a slab whose passes take 5 us each (ring of two C2
    // slabs on one device 125 against 200-230 us/step), against an all-reduce and an exchange of 20-30 us each that pass E
    // (>= 20 us) and the interior of pass A (>= 25 us) cover from ~150 k particles per slab.  Ranks may decide differently
    // (unequal column counts): both forms issue the same sequence of RCCL calls.
    // SPHX_SLAB_OVERLAP=always|never (read when a slab's buffers are made, not once per process: the tests run both forms).
    const char *ov = std::getenv("SPHX_SLAB_OVERLAP");
    const bool ov_always = ov && std::strcmp(ov, "always") == 0, ov_never = ov && std::strcmp(ov, "never") == 0;
    if (c->rebuild_every > 1 && !debug_switches().no_slab_overlap && !ov_never && (c->cap >= 150000 || ov_always)) {
        SPHX_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        SPHX_HIP(hipEventCreateWithFlags(&c->ev_cd, hipEventDisableTiming));
        SPHX_HIP(hipEventCreateWithFlags(&c->ev_ar, hipEventDisableTiming));
        SPHX_HIP(hipEventCreateWithFlags(&c->ev_max, hipEventDisableTiming));
        SPHX_HIP(hipEventCreateWithFlags(&c->ev_p, hipEventDisableTiming));
        SPHX_HIP(hipEventCreateWithFlags(&c->ev_u, hipEventDisableTiming));
        c->max_part.alloc(2 * kSlabMaxBlocks); c->max_part.zero(c->stream);
        c->ticket2.alloc(1); c->ticket2.zero(c->stream);
    }
    SPHX_HIP(hipStreamSynchronize(c->stream));
}

// End a capture that went wrong and drop whatever it produced.
void abandon_capture(hipStream_t st)
{
    hipGraph_t junk = nullptr;
    (void)hipStreamEndCapture(st, &junk);
    if (junk) (void)hipGraphDestroy(junk);
    (void)hipGetLastError();
}

struct PtrList {
    const double *p[16];
};
// max of one double per rank (single-process ring: stands in for the all-reduce)
__global__ void k_max_of(int n, PtrList src, double *out)  // thread j: component j of {max |v|, max drift}
{
    const int j = threadIdx.x;
    double m = src.p[0][j];
    for (int k = 1; k < n; ++k) m = fmax(m, src.p[k][j]);
    out[j] = m;
}

}  // namespace

SPHX_EXPORT int sphx_comm_available(void)
{
    SPHX_TRY
    (void)Rccl::get();  // loads librccl and resolves every entry point the native loop uses, or says what is missing
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_comm_unique_id(void *id_bytes, int capacity)
{
    SPHX_TRY
    require(id_bytes != nullptr && capacity >= NCCL_UNIQUE_ID_BYTES, "SPHX:Slab:rccl", "id buffer must hold 128 bytes");
    ncclUniqueId id;
    Rccl &R = Rccl::get();
    R.check(R.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(id_bytes, id.internal, NCCL_UNIQUE_ID_BYTES);
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_slab_comm_init(sphx_ctx *c, const void *id_bytes)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab && id_bytes != nullptr, "SPHX:Slab:ctx", "not a slab context");
    require(c->comm == nullptr, "SPHX:Slab:rccl", "communicator already initialised");
    ncclUniqueId id;
    std::memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
    Rccl &R = Rccl::get();
    slab_native_buffers(c);  // (before the collective call: a rank that cannot allocate must not leave the others inside it)
    SPHX_HIP(hipStreamSynchronize(c->stream));
    ncclComm_t comm = nullptr;
    R.check(R.CommInitRank(&comm, c->n_ranks, id, c->rank), "ncclCommInitRank");
    c->comm = comm;
    return SPHX_OK;
    SPHX_CATCH
}

// The exchange patterns of sphx_slab_run on a communicator of ONE rank (its own neighbour on both sides): two sends and two
// receives in one group, then four and four with both element types (a skinned step) -- the receives must be served in the
// order of the sends, which is what the two-rank ring relies on -- then the 16-byte all-reduce(max).  Checks the dlopen'ed entry points, enum values and stream use on this machine.
SPHX_EXPORT int sphx_comm_selftest(void)
{
    SPHX_TRY
    ensure_device();
    Rccl &R = Rccl::get();
    ncclUniqueId id;
    R.check(R.GetUniqueId(&id), "ncclGetUniqueId");
    ncclComm_t comm = nullptr;
    R.check(R.CommInitRank(&comm, 1, id, 0), "ncclCommInitRank");
    hipStream_t st = nullptr;
    const size_t n = 1000;
    DevBuf<double> sl(n), sr(n), rl(n), rr(n), v(2), vg(2);
    DevBuf<int> il(n), ir(n), jl(n), jr(n);
    std::vector<double> hl(n), hr(n), got_l(n), got_r(n);
    std::vector<int> kl(n), kr(n), gi_l(n), gi_r(n);
    for (size_t k = 0; k < n; ++k) { hl[k] = 1.0 + k; hr[k] = -2.0 - k; kl[k] = 7 + (int)k; kr[k] = -9 - (int)k; }
    const double hv[2] = {3.25, -1.5};
    double gv[2] = {0.0, 0.0};
    bool ok = false;
    try {
        SPHX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        sl.upload(hl.data(), n, st); sr.upload(hr.data(), n, st); il.upload(kl.data(), n, st); ir.upload(kr.data(), n, st);
        v.upload(hv, 2, st);
        rl.zero(st); rr.zero(st); jl.zero(st); jr.zero(st); vg.zero(st);
        auto ring = [&](const void *to_l, const void *to_r, void *from_l, void *from_r, ncclDataType_t ty) {
            R.check(R.GroupStart(), "ncclGroupStart");
            R.check(R.Send(to_l, n, ty, 0, comm, st), "ncclSend");
            R.check(R.Send(to_r, n, ty, 0, comm, st), "ncclSend");
            R.check(R.Recv(from_r, n, ty, 0, comm, st), "ncclRecv");
            R.check(R.Recv(from_l, n, ty, 0, comm, st), "ncclRecv");
            R.check(R.GroupEnd(), "ncclGroupEnd");
        };
        ring(sl.get(), sr.get(), rl.get(), rr.get(), ncclDouble);
        ring(il.get(), ir.get(), jl.get(), jr.get(), ncclInt32);
        SPHX_HIP(hipMemcpyAsync(got_l.data(), rl.get(), n * sizeof(double), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipMemcpyAsync(got_r.data(), rr.get(), n * sizeof(double), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipMemcpyAsync(gi_l.data(), jl.get(), n * sizeof(int), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipMemcpyAsync(gi_r.data(), jr.get(), n * sizeof(int), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipStreamSynchronize(st));
        const bool ok_two = got_r == hl && got_l == hr && gi_r == kl && gi_l == kr;
        // ... and the exchange of a skinned step: both messages and both id lists in ONE group (four sends, four receives)
        rl.zero(st); rr.zero(st); jl.zero(st); jr.zero(st);
        R.check(R.GroupStart(), "ncclGroupStart");
        R.check(R.Send(sl.get(), n, ncclDouble, 0, comm, st), "ncclSend");
        R.check(R.Send(sr.get(), n, ncclDouble, 0, comm, st), "ncclSend");
        R.check(R.Send(il.get(), n, ncclInt32, 0, comm, st), "ncclSend");
        R.check(R.Send(ir.get(), n, ncclInt32, 0, comm, st), "ncclSend");
        R.check(R.Recv(rr.get(), n, ncclDouble, 0, comm, st), "ncclRecv");
        R.check(R.Recv(rl.get(), n, ncclDouble, 0, comm, st), "ncclRecv");
        R.check(R.Recv(jr.get(), n, ncclInt32, 0, comm, st), "ncclRecv");
        R.check(R.Recv(jl.get(), n, ncclInt32, 0, comm, st), "ncclRecv");
        R.check(R.GroupEnd(), "ncclGroupEnd");
        R.check(R.AllReduce(v.get(), vg.get(), 2, ncclDouble, ncclMax, comm, st), "ncclAllReduce");
        SPHX_HIP(hipMemcpyAsync(got_l.data(), rl.get(), n * sizeof(double), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipMemcpyAsync(got_r.data(), rr.get(), n * sizeof(double), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipMemcpyAsync(gi_l.data(), jl.get(), n * sizeof(int), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipMemcpyAsync(gi_r.data(), jr.get(), n * sizeof(int), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipMemcpyAsync(gv, vg.get(), 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        SPHX_HIP(hipStreamSynchronize(st));
        // what went to the left arrives "from the right" and vice versa
        ok = ok_two && got_r == hl && got_l == hr && gi_r == kl && gi_l == kr && gv[0] == hv[0] && gv[1] == hv[1];
    } catch (...) {
        (void)R.CommDestroy(comm);
        if (st) (void)hipStreamDestroy(st);
        throw;
    }
    R.check(R.CommDestroy(comm), "ncclCommDestroy");
    SPHX_HIP(hipStreamDestroy(st));
    if (!ok) throw Error(SPHX_ERR_DEVICE, "SPHX:Slab:rccl", "RCCL self-test: the exchanged data came back wrong");
    return SPHX_OK;
    SPHX_CATCH
}

// The exchange of a skinned step and the all-reduce once more, this time CAPTURED into a hipGraph and replayed twice with
// fresh payloads on a communicator of one rank: tells whether this RCCL can be captured at all (sphx_slab_graph_prepare
// relies on it) before a multi-rank run stakes its loop on it.
SPHX_EXPORT int sphx_comm_selftest_graph(void)
{
    SPHX_TRY
    ensure_device();
    Rccl &R = Rccl::get();
    ncclUniqueId id;
    R.check(R.GetUniqueId(&id), "ncclGetUniqueId");
    ncclComm_t comm = nullptr;
    R.check(R.CommInitRank(&comm, 1, id, 0), "ncclCommInitRank");
    hipStream_t st = nullptr;
    hipGraph_t g = nullptr;
    hipGraphExec_t exec = nullptr;
    const size_t n = 1000;
    DevBuf<double> sl(n), sr(n), rl(n), rr(n), v(2), vg(2);
    DevBuf<int> il(n), ir(n), jl(n), jr(n);
    std::vector<double> hl(n), hr(n), got_l(n), got_r(n);
    std::vector<int> kl(n), kr(n), gi_l(n), gi_r(n);
    bool ok = true;
    auto exchange = [&]() {
        R.check(R.AllReduce(v.get(), vg.get(), 2, ncclDouble, ncclMax, comm, st), "ncclAllReduce");
        R.check(R.GroupStart(), "ncclGroupStart");
        R.check(R.Send(sl.get(), n, ncclDouble, 0, comm, st), "ncclSend");
        R.check(R.Send(sr.get(), n, ncclDouble, 0, comm, st), "ncclSend");
        R.check(R.Send(il.get(), n, ncclInt32, 0, comm, st), "ncclSend");
        R.check(R.Send(ir.get(), n, ncclInt32, 0, comm, st), "ncclSend");
        R.check(R.Recv(rr.get(), n, ncclDouble, 0, comm, st), "ncclRecv");
        R.check(R.Recv(rl.get(), n, ncclDouble, 0, comm, st), "ncclRecv");
        R.check(R.Recv(jr.get(), n, ncclInt32, 0, comm, st), "ncclRecv");
        R.check(R.Recv(jl.get(), n, ncclInt32, 0, comm, st), "ncclRecv");
        R.check(R.GroupEnd(), "ncclGroupEnd");
    };
    try {
        SPHX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        rl.zero(st); rr.zero(st); jl.zero(st); jr.zero(st); vg.zero(st); v.zero(st); sl.zero(st); sr.zero(st); il.zero(st); ir.zero(st);
        exchange();  // eagerly first: channels and buffers of the communicator are set up outside the capture
        SPHX_HIP(hipStreamSynchronize(st));
        SPHX_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        try {
            exchange();
            exchange();  // (two steps per graph, as the step graph holds several)
        } catch (...) {
            abandon_capture(st);
            throw;
        }
        SPHX_HIP(hipStreamEndCapture(st, &g));
        SPHX_HIP(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2 && ok; ++rep) {
            for (size_t k = 0; k < n; ++k) {
                hl[k] = 1.0 + k + 1000.0 * rep; hr[k] = -2.0 - k - 1000.0 * rep;
                kl[k] = 7 + (int)k + 100000 * rep; kr[k] = -9 - (int)k - 100000 * rep;
            }
            const double hv[2] = {3.25 + rep, -1.5 - rep};
            double gv[2] = {0.0, 0.0};
            sl.upload(hl.data(), n, st); sr.upload(hr.data(), n, st); il.upload(kl.data(), n, st); ir.upload(kr.data(), n, st);
            v.upload(hv, 2, st);
            rl.zero(st); rr.zero(st); jl.zero(st); jr.zero(st); vg.zero(st);
            SPHX_HIP(hipStreamSynchronize(st));  // (the host vectors are reused)
            SPHX_HIP(hipGraphLaunch(exec, st));
            SPHX_HIP(hipMemcpyAsync(got_l.data(), rl.get(), n * sizeof(double), hipMemcpyDeviceToHost, st));
            SPHX_HIP(hipMemcpyAsync(got_r.data(), rr.get(), n * sizeof(double), hipMemcpyDeviceToHost, st));
            SPHX_HIP(hipMemcpyAsync(gi_l.data(), jl.get(), n * sizeof(int), hipMemcpyDeviceToHost, st));
            SPHX_HIP(hipMemcpyAsync(gi_r.data(), jr.get(), n * sizeof(int), hipMemcpyDeviceToHost, st));
            SPHX_HIP(hipMemcpyAsync(gv, vg.get(), 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            SPHX_HIP(hipStreamSynchronize(st));
            ok = got_r == hl && got_l == hr && gi_r == kl && gi_l == kr && gv[0] == hv[0] && gv[1] == hv[1];
        }
    } catch (...) {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (g) (void)hipGraphDestroy(g);
        (void)R.CommDestroy(comm);
        if (st) (void)hipStreamDestroy(st);
        throw;
    }
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(g);
    R.check(R.CommDestroy(comm), "ncclCommDestroy");
    SPHX_HIP(hipStreamDestroy(st));
    if (!ok) throw Error(SPHX_ERR_DEVICE, "SPHX:Slab:rccl", "RCCL graph self-test: a replayed exchange delivered the wrong data");
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_slab_comm_destroy(sphx_ctx *c)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab, "SPHX:Slab:ctx", "not a slab context");
    if (c->comm) {
        SPHX_HIP(hipStreamSynchronize(c->stream));
        Rccl &R = Rccl::get();
        R.check(R.CommDestroy(c->comm), "ncclCommDestroy");
        c->comm = nullptr;
    }
    return SPHX_OK;
    SPHX_CATCH
}

namespace {

// ---- one step of a skinned slab, in four phases separated by the three exchanges (see SlabLists) ----
FluidSet slab_new_state(sphx_ctx *c) { return c->view(1 - c->cur, 0); }  // S[1-q] with the (in-place) layout arrays

constexpr int kTicketBlocks = 1024;  // largest grid of the kernels that end with last_workgroup_out (~20 ns per ticket)

void slab_phase1(sphx_ctx *c)  // passes A..E into S[1-q]; the tail workgroup of pass E leaves the local maxima in vmax_l[0..1]
{
    const int q = c->cur;
    const FluidSet s = c->view(q, 0), o = c->view(1 - q, 0);
    FluidTmp t = c->tmp;
    t.posn = o.pos; t.veln = o.vel; t.drhon = o.drho;
    t.seal_out = c->vmax_l.get();
    launch_physics_any(c, q, s, t, 0, 0, 3, 2);
}

// ... in three pieces (contexts with a second stream, see sphx_ctx::stream2): passes A, B, CD on the slab's stream; the local
// maxima -- both exist once pass CD is through -- on `aux`, where the all-reduce follows them; pass E, without a tail, on the
// slab's stream again, beside the two
// pass A of the step slot of parity q on S[q]: dmode 3 = the cell sweep (fresh grid only) and the walk of `part` (0 all,
// 1 interior, 2 boundary workgroups, see slab_part_skips); dmode 4 = that walk alone
void slab_pass_a(sphx_ctx *c, int q, int dmode, int part)
{
    const FluidSet s = c->view(q, 0), o = c->view(1 - q, 0);
    FluidTmp t = c->tmp;
    t.posn = o.pos; t.veln = o.vel; t.drhon = o.drho;
    c->pass_a_part = part;
    launch_physics_any(c, q, s, t, 0, 1, dmode);
    c->pass_a_part = 0;
}
void slab_phase1_abc(sphx_ctx *c)
{
    const int q = c->cur;
    const FluidSet s = c->view(q, 0), o = c->view(1 - q, 0);
    FluidTmp t = c->tmp;
    t.posn = o.pos; t.veln = o.vel; t.drhon = o.drho;
    // (the interior workgroups of this pass A may have run already, behind the previous step's pack3: slab_pass_a_interior)
    slab_pass_a(c, q, 3, c->a_interior_done ? 2 : 0);
    c->a_interior_done = false;
    launch_physics_any(c, q, s, t, 0, 2);
    launch_physics_any(c, q, s, t, 0, 3);
}
// message A of the step just taken is packed, the clock advanced, the host's parity flipped (slab_phase4): the interior
// workgroups of the NEXT step's pass A need nothing the exchange brings -- they run while it is under way
void slab_pass_a_interior(sphx_ctx *c)
{
    slab_pass_a(c, c->cur, 4, 1);
    c->a_interior_done = true;
}
void slab_local_maxima_of_step(sphx_ctx *c, hipStream_t aux)
{
    const int q = c->cur;
    const FluidSet s = c->view(q, 0), o = c->view(1 - q, 0);
    hipLaunchKernelGGL(k_slab_maxima, dim3(kSlabMaxBlocks), dim3(kBlock), 0, aux, (const Clock *)c->clock.get(), q, c->grid,
                       c->n_vpart, (const double *)c->dpart.get(), (const double2 *)o.vel, (const int *)s.cell, c->max_part.get(),
                       c->vmax_l.get(), c->ticket2.get());
}
void slab_phase1_e(sphx_ctx *c)
{
    const int q = c->cur;
    const FluidSet s = c->view(q, 0), o = c->view(1 - q, 0);
    FluidTmp t = c->tmp;
    t.posn = o.pos; t.veln = o.vel; t.drhon = o.drho;
    launch_physics_any(c, q, s, t, 0, 4);
}

void slab_phase2(sphx_ctx *c)  // global maxima known: re-binning decision, message A, clock -- one launch
{
    const int q = c->cur;
    SlabPack p = c->pack;
    p.send_l = c->msg_sl.get();
    p.send_r = c->msg_sr.get();
    // grid-stride on at most one workgroup per CU: the kernel ends with a ticket per workgroup (last_workgroup_out)
    const unsigned nb = std::min<unsigned>(kTicketBlocks, std::max<unsigned>(c->n_blocks_flat, div_up((size_t)2 * c->msg_cap, kBlock)));
    launch(c, "k_slab_pack3", k_slab_pack3, dim3(nb), dim3(kBlock), c->clock.get(), q, c->grid, c->phys, slab_new_state(c), p,
           c->lists, (const double *)c->vmax_g.get(), c->flags.get(), c->half_skin(), c->rebuild_every, c->ticket.get());
}

void slab_phase3(sphx_ctx *c)  // message A arrived (and with it the ids of the lists the neighbours made in the previous step):
                               // refresh the halo copies, or re-bin and make the lists of the next cycle
{
    const int q = c->cur, qf = q | kOnlyIfRebuild;
    Clock *clk = c->clock.get();
    const FluidSet d = slab_new_state(c);
    launch(c, "k_slab_unpack3", k_slab_unpack3, dim3(div_up((size_t)2 * c->msg_cap, kBlock)), dim3(kBlock), clk, q, c->grid, d,
           c->pack, c->lists, (const double *)c->msg_rl.get(), (const double *)c->msg_rr.get(), (const int *)c->ids_r_[0].get(),
           (const int *)c->ids_r_[1].get(), c->n_new.get(), c->flags.get(), c->ticket.get());
    const int kRebinBlocks = 4096;  // grid-stride: on the steps that do not re-bin these launches return at once
    launch_cell_scan(c, clk, qf, d.start);
    launch_scatter_reorder(c, clk, qf,
                           reorder_args(c->kpos.get(), c->kvel.get(), c->kdrho.get(), c->kmass.get(), c->kid.get(), d, nullptr,
                                        c->slot_of_id.get()),
                           d, kRebinBlocks);
    launch(c, "k_slab_sendlist", k_slab_sendlist, dim3(std::min(c->n_blocks_flat, kTicketBlocks)), dim3(kBlock), (const Clock *)clk,
           q, c->grid, d, c->pack, c->lists, c->flags.get(), 0, c->ticket.get());
}

void slab_phase4(sphx_ctx *c)  // host bookkeeping of the step (the ids made in phase 3 travel with the next step's message A)
{
    SPHX_HIP(hipGetLastError());
    c->cur ^= 1;
    c->slab_steps_enqueued += 1;
}

// the exchange lists of the first cycle (the layout of slab_setup): send lists + ids out, then (ids in) receive slots
void slab_lists_out(sphx_ctx *c)
{
    const FluidSet d = c->view(c->cur, 0);
    SPHX_HIP(hipMemsetAsync(c->send_cnt.get(), 0, 2 * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_slab_sendlist, dim3(std::min(c->n_blocks_flat, kTicketBlocks)), dim3(kBlock), 0, c->stream,
                       (const Clock *)c->clock.get(), 0, c->grid, d, c->pack, c->lists, c->flags.get(), 1, c->ticket.get());
}
void slab_lists_in(sphx_ctx *c)
{
    hipLaunchKernelGGL(k_slab_recvslots, dim3(div_up((size_t)2 * c->msg_cap, kBlock)), dim3(kBlock), 0, c->stream,
                       (const Clock *)c->clock.get(), 0, c->pack, c->lists, (const int *)c->ids_r_[0].get(),
                       (const int *)c->ids_r_[1].get(), (const int *)c->view(c->cur, 0).id, c->flags.get(), 1);
    SPHX_HIP(hipGetLastError());
    c->lists_ready = true;
}

void slab_local_maxima(sphx_ctx *c)  // arming: max |v| of the owned particles of the current state, drift 0
{
    const FluidSet fs = c->view(c->cur, c->rebuild_every > 1 ? 0 : c->cur);
    SPHX_HIP(hipMemsetAsync(c->vmax_l.get(), 0, 2 * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_vmax_init, dim3(1), dim3(kScanBlock), 0, c->stream, c->clock.get(), c->grid, (const double2 *)fs.pos,
                       (const double2 *)fs.vel, c->vmax_l.get(), (const int *)fs.cell);
}

}  // namespace

namespace {

constexpr int kSlabGraphSteps = 10;  // steps per replay of a slab's step graph (even: the state parity comes back)

// ---- one rank per process: the steps of sphx_slab_run on the context's stream (eagerly, or under stream capture) ----
struct RcclLoop {
    sphx_ctx *c;
    Rccl &R;
    hipStream_t st;
    int left, right;
    size_t n_msg, n_ids;
    explicit RcclLoop(sphx_ctx *ctx)
        : c(ctx), R(Rccl::get()), st(ctx->stream), left((ctx->rank + ctx->n_ranks - 1) % ctx->n_ranks),
          right((ctx->rank + 1) % ctx->n_ranks), n_msg(1 + 7 * (size_t)ctx->msg_cap), n_ids(1 + (size_t)ctx->msg_cap) {}
    // My left message is my left neighbour's "from the right" message and vice versa; with two ranks both go to the same
    // peer, which posts its receives in the order the sends are posted here.
    void ring(const void *to_l, const void *to_r, void *from_l, void *from_r, size_t count, ncclDataType_t ty)
    {
        R.check(R.GroupStart(), "ncclGroupStart");
        R.check(R.Send(to_l, count, ty, left, c->comm, st), "ncclSend");
        R.check(R.Send(to_r, count, ty, right, c->comm, st), "ncclSend");
        R.check(R.Recv(from_r, count, ty, right, c->comm, st), "ncclRecv");
        R.check(R.Recv(from_l, count, ty, left, c->comm, st), "ncclRecv");
        R.check(R.GroupEnd(), "ncclGroupEnd");
    }
    // the exchange of a skinned step: message A and the list ids of the previous step in ONE group
    void ring_step()
    {
        R.check(R.GroupStart(), "ncclGroupStart");
        R.check(R.Send(c->msg_sl.get(), n_msg, ncclDouble, left, c->comm, st), "ncclSend");
        R.check(R.Send(c->msg_sr.get(), n_msg, ncclDouble, right, c->comm, st), "ncclSend");
        R.check(R.Send(c->ids_s_[0].get(), n_ids, ncclInt32, left, c->comm, st), "ncclSend");
        R.check(R.Send(c->ids_s_[1].get(), n_ids, ncclInt32, right, c->comm, st), "ncclSend");
        R.check(R.Recv(c->msg_rr.get(), n_msg, ncclDouble, right, c->comm, st), "ncclRecv");
        R.check(R.Recv(c->msg_rl.get(), n_msg, ncclDouble, left, c->comm, st), "ncclRecv");
        R.check(R.Recv(c->ids_r_[1].get(), n_ids, ncclInt32, right, c->comm, st), "ncclRecv");
        R.check(R.Recv(c->ids_r_[0].get(), n_ids, ncclInt32, left, c->comm, st), "ncclRecv");
        R.check(R.GroupEnd(), "ncclGroupEnd");
    }
    void step()
    {
        double *vl = c->vmax_l.get(), *vg = c->vmax_g.get();
        if (c->rebuild_every <= 1) {
            slab_compute_impl(c, c->msg_sl.get(), c->msg_sr.get(), vl);
            R.check(R.AllReduce(vl, vg, 1, ncclDouble, ncclMax, c->comm, st), "ncclAllReduce");
            ring(c->msg_sl.get(), c->msg_sr.get(), c->msg_rl.get(), c->msg_rr.get(), n_msg, ncclDouble);
            slab_finish_impl(c, c->msg_rl.get(), c->msg_rr.get(), vg);
        } else if (c->stream2 && !serial_aux) {
            // the maxima and the all-reduce on the second stream, beside pass E; then the exchange and what follows it on the
            // second stream, beside the interior workgroups of the next step's pass A
            slab_phase1_abc(c);
            SPHX_HIP(hipEventRecord(c->ev_cd, st));
            SPHX_HIP(hipStreamWaitEvent(c->stream2, c->ev_cd, 0));
            slab_local_maxima_of_step(c, c->stream2);
            R.check(R.AllReduce(vl, vg, 2, ncclDouble, ncclMax, c->comm, c->stream2), "ncclAllReduce");
            SPHX_HIP(hipEventRecord(c->ev_ar, c->stream2));
            slab_phase1_e(c);
            SPHX_HIP(hipStreamWaitEvent(st, c->ev_ar, 0));
            slab_phase2(c);
            SPHX_HIP(hipEventRecord(c->ev_p, st));
            SPHX_HIP(hipStreamWaitEvent(c->stream2, c->ev_p, 0));
            {
                hipStream_t keep = st;
                st = c->stream2; c->stream = c->stream2;  // (ring_step and slab_phase3 enqueue on "the" stream)
                try { ring_step(); slab_phase3(c); } catch (...) { st = keep; c->stream = keep; throw; }
                st = keep; c->stream = keep;
            }
            SPHX_HIP(hipEventRecord(c->ev_u, c->stream2));
            slab_phase4(c);
            slab_pass_a_interior(c);
            SPHX_HIP(hipStreamWaitEvent(st, c->ev_u, 0));
        } else if (c->stream2) {  // (under stream capture: the same pieces in one chain)
            slab_phase1_abc(c);
            slab_local_maxima_of_step(c, st);
            R.check(R.AllReduce(vl, vg, 2, ncclDouble, ncclMax, c->comm, st), "ncclAllReduce");
            slab_phase1_e(c);
            slab_phase2(c);
            ring_step();
            slab_phase3(c);
            slab_phase4(c);
        } else {
            slab_phase1(c);
            R.check(R.AllReduce(vl, vg, 2, ncclDouble, ncclMax, c->comm, st), "ncclAllReduce");
            slab_phase2(c);
            ring_step();
            slab_phase3(c);
            slab_phase4(c);
        }
    }
    bool serial_aux = false;  // true under stream capture: everything on the one capturing stream
};

// ---- all slabs of the ring in one process: the same steps with device-to-device copies and events ----
struct GroupLoop {
    sphx_ctx **ctxs;
    int n;
    PtrList vls{};
    bool skinned;
    bool serial = false;  // every slab enqueues on the same stream: stream order is the only ordering needed
    size_t msg_bytes, ids_bytes;
    GroupLoop(sphx_ctx **cs, int count) : ctxs(cs), n(count)
    {
        for (int r = 0; r < n; ++r) vls.p[r] = ctxs[r]->vmax_l.get();
        skinned = ctxs[0]->rebuild_every > 1;
        msg_bytes = (1 + 7 * (size_t)ctxs[0]->msg_cap) * sizeof(double);
        ids_bytes = (1 + (size_t)ctxs[0]->msg_cap) * sizeof(int);
    }
    // A phase boundary of the ring: every rank records "my outputs of this phase are complete" and, before it touches
    // anything another rank produced (or overwrites what another rank may still be reading), waits for all the others.
    void done(hipEvent_t sphx_ctx::*ev)
    {
        if (serial) return;
        for (int r = 0; r < n; ++r) SPHX_HIP(hipEventRecord(ctxs[r]->*ev, ctxs[r]->stream));
    }
    void wait_others(int r, hipEvent_t sphx_ctx::*ev)
    {
        if (serial) return;
        for (int o = 0; o < n; ++o) if (o != r) SPHX_HIP(hipStreamWaitEvent(ctxs[r]->stream, ctxs[o]->*ev, 0));
    }
    void max_of_all(sphx_ctx *c) { hipLaunchKernelGGL(k_max_of, dim3(1), dim3(2), 0, c->stream, n, vls, c->vmax_g.get()); }
    void copy_msgs(int r)  // what my ring neighbours addressed to me
    {
        sphx_ctx *c = ctxs[r], *L = ctxs[(r + n - 1) % n], *Rr = ctxs[(r + 1) % n];
        SPHX_HIP(hipMemcpyAsync(c->msg_rl.get(), L->msg_sr.get(), msg_bytes, hipMemcpyDeviceToDevice, c->stream));
        SPHX_HIP(hipMemcpyAsync(c->msg_rr.get(), Rr->msg_sl.get(), msg_bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    void copy_ids(int r)
    {
        sphx_ctx *c = ctxs[r], *L = ctxs[(r + n - 1) % n], *Rr = ctxs[(r + 1) % n];
        SPHX_HIP(hipMemcpyAsync(c->ids_r_[0].get(), L->ids_s_[1].get(), ids_bytes, hipMemcpyDeviceToDevice, c->stream));
        SPHX_HIP(hipMemcpyAsync(c->ids_r_[1].get(), Rr->ids_s_[0].get(), ids_bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    // entry_waits: the slabs first wait until the others have consumed their previous messages and maxima (not the first
    // step of a captured graph: what came before is ordered by the launch, see replay())
    void step(bool entry_waits = true)
    {
        const bool overlap = skinned && ctxs[0]->stream2 != nullptr;
        if (overlap) {
            // passes A, B, CD; the local maxima on every slab's second stream (serial: on the one stream); pass E beside them;
            // the stand-in for the all-reduce on the second stream once everybody's maxima are out; join in front of pack3
            auto aux = [&](sphx_ctx *c) { return serial ? c->stream : c->stream2; };
            for (int r = 0; r < n; ++r) {
                sphx_ctx *c = ctxs[r];
                if (entry_waits) wait_others(r, &sphx_ctx::ev_received);
                slab_phase1_abc(c);
                if (!serial) {
                    SPHX_HIP(hipEventRecord(c->ev_cd, c->stream));
                    SPHX_HIP(hipStreamWaitEvent(c->stream2, c->ev_cd, 0));
                }
                slab_local_maxima_of_step(c, aux(c));
                if (!serial) SPHX_HIP(hipEventRecord(c->ev_max, c->stream2));
            }
            for (int r = 0; r < n; ++r) slab_phase1_e(ctxs[r]);
            for (int r = 0; r < n; ++r) {  // "all-reduce", decision, message A
                sphx_ctx *c = ctxs[r];
                if (!serial)
                    for (int o = 0; o < n; ++o) if (o != r) SPHX_HIP(hipStreamWaitEvent(c->stream2, ctxs[o]->ev_max, 0));
                hipLaunchKernelGGL(k_max_of, dim3(1), dim3(2), 0, aux(c), n, vls, c->vmax_g.get());
                if (!serial) {
                    SPHX_HIP(hipEventRecord(c->ev_ar, c->stream2));
                    SPHX_HIP(hipStreamWaitEvent(c->stream, c->ev_ar, 0));
                }
                slab_phase2(c);
                // "my maxima have been read by me, my message A is complete" -- recorded HERE, in front of the interior
                // workgroups of the next step's pass A, which the neighbours need not wait for (the in-process ring keeps the
                // copies that stand in for the exchange on the slab's own stream: what it tests is the split itself)
                if (!serial) SPHX_HIP(hipEventRecord(c->ev_received, c->stream));
                slab_pass_a(c, 1 - c->cur, 4, 1);
                c->a_interior_done = true;
            }
        } else {
        for (int r = 0; r < n; ++r) {
            sphx_ctx *c = ctxs[r];
            if (entry_waits) wait_others(r, &sphx_ctx::ev_received);
            if (skinned) slab_phase1(c);
            else slab_compute_impl(c, c->msg_sl.get(), c->msg_sr.get(), c->vmax_l.get());
        }
        done(&sphx_ctx::ev_computed);
        if (!skinned) {
            for (int r = 0; r < n; ++r) {
                sphx_ctx *c = ctxs[r];
                wait_others(r, &sphx_ctx::ev_computed);
                copy_msgs(r);
                max_of_all(c);
            }
            done(&sphx_ctx::ev_received);
            for (int r = 0; r < n; ++r) slab_finish_impl(ctxs[r], ctxs[r]->msg_rl.get(), ctxs[r]->msg_rr.get(), ctxs[r]->vmax_g.get());
            return;
        }
        for (int r = 0; r < n; ++r) {  // "all-reduce", decision, message A
            sphx_ctx *c = ctxs[r];
            wait_others(r, &sphx_ctx::ev_computed);
            max_of_all(c);
            slab_phase2(c);
        }
        }
        if (!overlap) done(&sphx_ctx::ev_received);  // (reused: "my maxima have been read by me, my message A is complete")
        for (int r = 0; r < n; ++r) {  // message A and the ids of the previous step's lists in
            wait_others(r, &sphx_ctx::ev_received);
            copy_msgs(r);
            copy_ids(r);
        }
        done(&sphx_ctx::ev_computed);  // (reused: "I have taken what the neighbours addressed to me")
        for (int r = 0; r < n; ++r) {  // halo refresh or re-binning chain; the lists / ids of the next cycle overwrite the old ones
            wait_others(r, &sphx_ctx::ev_computed);
            slab_phase3(ctxs[r]);
            slab_phase4(ctxs[r]);
        }
        done(&sphx_ctx::ev_received);
    }
    bool graph_matches() const
    {
        const sphx_ctx *c0 = ctxs[0];
        if (!c0->steps_graph || (int)c0->steps_graph_ring.size() != n || c0->steps_graph_cur != c0->cur) return false;
        for (int r = 0; r < n; ++r) if (c0->steps_graph_ring[r] != ctxs[r]) return false;
        return true;
    }
    // one replay of slab 0's step graph: everything the slabs enqueued before it is ordered ahead of it, everything they
    // enqueue afterwards behind it
    void replay()
    {
        sphx_ctx *c0 = ctxs[0];
        for (int r = 1; r < n; ++r) {
            SPHX_HIP(hipEventRecord(ctxs[r]->ev_join, ctxs[r]->stream));
            SPHX_HIP(hipStreamWaitEvent(c0->stream, ctxs[r]->ev_join, 0));
        }
        SPHX_HIP(hipGraphLaunch(c0->steps_graph, c0->stream));
        SPHX_HIP(hipEventRecord(c0->ev_fork, c0->stream));
        for (int r = 1; r < n; ++r) SPHX_HIP(hipStreamWaitEvent(ctxs[r]->stream, c0->ev_fork, 0));
        done(&sphx_ctx::ev_received);  // (the eager steps that may follow wait for these)
        for (int r = 0; r < n; ++r) { ctxs[r]->slab_steps_enqueued += kSlabGraphSteps; ctxs[r]->a_interior_done = false; }
    }
};

void check_ring(sphx_ctx **ctxs, int n)
{
    require(ctxs != nullptr && n >= 2 && n <= 16, "SPHX:Slab:group", "a ring needs 2..16 slab contexts");
    for (int r = 0; r < n; ++r) {
        sphx_ctx *c = ctxs[r];
        require(c != nullptr && c->is_slab && c->rank == r && c->n_ranks == n, "SPHX:Slab:group",
                "ctxs[r] must be slab r of an n-slab ring");
        require(c->msg_cap == ctxs[0]->msg_cap && c->rebuild_every == ctxs[0]->rebuild_every, "SPHX:Slab:group",
                "slabs of one ring share the message capacity and the re-binning interval");
        slab_native_buffers(c);
        if (!c->ev_computed) {
            SPHX_HIP(hipEventCreateWithFlags(&c->ev_computed, hipEventDisableTiming));
            SPHX_HIP(hipEventCreateWithFlags(&c->ev_received, hipEventDisableTiming));
            SPHX_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            SPHX_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
    }
}

}  // namespace

SPHX_EXPORT int sphx_slab_run(sphx_ctx *c, double t_target, int64_t n_steps)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab, "SPHX:Slab:ctx", "not a slab context");
    require(c->comm != nullptr, "SPHX:Slab:rccl", "sphx_slab_comm_init first");
    require(n_steps > 0, "SPHX:Ctx:steps", "n_steps must be positive");
    RcclLoop loop(c);
    double *vl = c->vmax_l.get(), *vg = c->vmax_g.get();
    const bool skinned = c->rebuild_every > 1;
    c->a_interior_done = false;  // (whatever the previous call launched ahead found the loop stopped: the first step sweeps / walks everything)
    // arm the clock with the global max |v| of the current state: after a step the clock holds it already (the all-reduced
    // value that step's dt rule used); only a state that has never been stepped needs the reduction (k_vmax_init is one
    // workgroup over the whole slab: 0.9 ms at 0.76 M particles)
    if (c->slab_steps_enqueued == 0) {
        slab_local_maxima(c);
        loop.R.check(loop.R.AllReduce(vl, vg, 2, ncclDouble, ncclMax, c->comm, loop.st), "ncclAllReduce");
        arm_clock(c, t_target, (long long)n_steps, c->cur, (const double *)vg);
    } else {
        arm_clock(c, t_target, (long long)n_steps, c->cur, (const double *)nullptr);
    }
    if (skinned && !c->lists_ready) {
        slab_lists_out(c);
        loop.ring(c->ids_s_[0].get(), c->ids_s_[1].get(), c->ids_r_[0].get(), c->ids_r_[1].get(), loop.n_ids, ncclInt32);
        slab_lists_in(c);
    }
    int64_t k = 0;
    if (c->steps_graph && c->steps_graph_cur == c->cur)  // whole replays of the step graph (sphx_slab_graph_prepare) ...
        for (; n_steps - k >= kSlabGraphSteps; k += kSlabGraphSteps) {
            SPHX_HIP(hipGraphLaunch(c->steps_graph, loop.st));
            c->slab_steps_enqueued += kSlabGraphSteps;
            c->a_interior_done = false;
        }
    for (; k < n_steps; ++k) loop.step();  // ... the rest step by step
    SPHX_HIP(hipGetLastError());
    return SPHX_OK;
    SPHX_CATCH
}

SPHX_EXPORT int sphx_slab_group_run(sphx_ctx **ctxs, int n, double t_target, int64_t n_steps)
{
    SPHX_TRY
    check_ring(ctxs, n);
    require(n_steps > 0, "SPHX:Ctx:steps", "n_steps must be positive");
    GroupLoop loop(ctxs, n);
    for (int r = 0; r < n; ++r) ctxs[r]->a_interior_done = false;  // (see sphx_slab_run)
    // arm: local maxima -> global -> clock (a state that has been stepped: the clock holds the global maximum already, see
    // sphx_slab_run); first exchange lists
    bool fresh_state = false;
    for (int r = 0; r < n; ++r) fresh_state = fresh_state || ctxs[r]->slab_steps_enqueued == 0;
    if (fresh_state)
        for (int r = 0; r < n; ++r) slab_local_maxima(ctxs[r]);
    loop.done(&sphx_ctx::ev_computed);
    for (int r = 0; r < n; ++r) {
        sphx_ctx *c = ctxs[r];
        loop.wait_others(r, &sphx_ctx::ev_computed);
        if (fresh_state) loop.max_of_all(c);
        arm_clock(c, t_target, (long long)n_steps, c->cur, fresh_state ? (const double *)c->vmax_g.get() : (const double *)nullptr);
        if (loop.skinned && !c->lists_ready) slab_lists_out(c);
    }
    loop.done(&sphx_ctx::ev_received);
    if (loop.skinned && !ctxs[0]->lists_ready) {
        for (int r = 0; r < n; ++r) {
            loop.wait_others(r, &sphx_ctx::ev_received);
            loop.copy_ids(r);
            slab_lists_in(ctxs[r]);
        }
        loop.done(&sphx_ctx::ev_received);
    }
    int64_t k = 0;
    if (loop.graph_matches())
        for (; n_steps - k >= kSlabGraphSteps; k += kSlabGraphSteps) loop.replay();
    for (; k < n_steps; ++k) loop.step();
    SPHX_HIP(hipGetLastError());
    return SPHX_OK;
    SPHX_CATCH
}

// Capture kSlabGraphSteps whole steps of the native loop into one graph; sphx_slab_run / sphx_slab_group_run then replay it
// for every full batch of that many steps and launch only the remainder step by step.  n = 1: the context of this
// process's rank (sphx_slab_run; the RCCL calls are captured with the kernels -- every rank of the communicator must
// make this call at the same point of its sequence, because the graph is warmed with one idle replay that exchanges
// messages); n >= 2: an in-process ring (the graph is held by slab 0).  The loop must have run at least two steps
// (communicator channels, exchange lists and pools are set up by then: nothing may allocate during capture).  Without
// this call the loops stay eager; a failed capture leaves them eager and reports the error.
SPHX_EXPORT int sphx_slab_graph_prepare(sphx_ctx **ctxs, int n)
{
    SPHX_TRY
    require(ctxs != nullptr && n >= 1 && ctxs[0] != nullptr && ctxs[0]->is_slab, "SPHX:Slab:ctx", "not a slab context");
    sphx_ctx *c0 = ctxs[0];
    require(c0->rebuild_every > 1, "SPHX:Slab:protocol", "step graphs are for skinned slabs (rebuild_every != 1)");
    if (n == 1) require(c0->comm != nullptr, "SPHX:Slab:rccl", "sphx_slab_comm_init first");
    else check_ring(ctxs, n);
    for (int r = 0; r < n; ++r) {
        require(ctxs[r]->lists_ready && ctxs[r]->slab_steps_enqueued >= 2, "SPHX:Slab:graph",
                "run at least two steps before capturing the step graph");
        SPHX_HIP(hipStreamSynchronize(ctxs[r]->stream));
    }
    if (c0->steps_graph) { (void)hipGraphExecDestroy(c0->steps_graph); c0->steps_graph = nullptr; }
    const int64_t enq0 = c0->slab_steps_enqueued;
    const int cur0 = c0->cur;
    hipStream_t s0 = c0->stream;
    hipGraph_t g = nullptr;
    // An in-process ring is captured as ONE chain on slab 0's stream (the slabs take turns phase by phase): a graph whose
    // branches meet at every phase boundary replays slower than the eager loop on ROCm 7.2 (0.25 M particles per slab: 334
    // against 289 us/step), and the slabs of a ring share one device anyway.
    std::vector<hipStream_t> own_streams(n);
    for (int r = 0; r < n; ++r) { own_streams[r] = ctxs[r]->stream; ctxs[r]->stream = s0; }
    auto restore_streams = [&]() { for (int r = 0; r < n; ++r) ctxs[r]->stream = own_streams[r]; };
    for (int r = 0; r < n; ++r) ctxs[r]->a_interior_done = false;  // (the graph's first step sweeps / walks everything)
    const hipError_t e_begin = hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed);
    if (e_begin != hipSuccess) { restore_streams(); SPHX_HIP(e_begin); }
    try {
        if (n == 1) {
            RcclLoop loop(c0);
            loop.serial_aux = true;
            for (int k = 0; k < kSlabGraphSteps; ++k) loop.step();
        } else {
            GroupLoop loop(ctxs, n);
            loop.serial = true;
            for (int k = 0; k < kSlabGraphSteps; ++k) loop.step();
        }
    } catch (...) {
        abandon_capture(s0);
        restore_streams();
        for (int r = 0; r < n; ++r) { ctxs[r]->cur = cur0; ctxs[r]->slab_steps_enqueued = enq0; }
        throw;
    }
    restore_streams();
    for (int r = 0; r < n; ++r) { ctxs[r]->slab_steps_enqueued -= kSlabGraphSteps; ctxs[r]->a_interior_done = false; }  // nothing has executed
    const hipError_t e_end = hipStreamEndCapture(s0, &g);
    if (e_end != hipSuccess) { (void)hipGetLastError(); if (g) (void)hipGraphDestroy(g); SPHX_HIP(e_end); }
    hipGraphExec_t exec = nullptr;
    const hipError_t e_inst = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    SPHX_HIP(e_inst);
    // warm replay with the clock disarmed: every kernel returns at once, the messages that travel are ignored (ids made by
    // the last real step are delivered -- that is what the next real step would have done first)
    for (int r = 0; r < n; ++r) hipLaunchKernelGGL(k_disarm, dim3(1), dim3(1), 0, ctxs[r]->stream, ctxs[r]->clock.get());
    c0->steps_graph = exec;
    c0->steps_graph_cur = cur0;
    c0->steps_graph_ring.assign(ctxs, ctxs + n);
    if (n == 1) {
        SPHX_HIP(hipGraphLaunch(exec, s0));
    } else {
        GroupLoop loop(ctxs, n);
        loop.replay();
        for (int r = 0; r < n; ++r) ctxs[r]->slab_steps_enqueued -= kSlabGraphSteps;
    }
    for (int r = 0; r < n; ++r) SPHX_HIP(hipStreamSynchronize(ctxs[r]->stream));
    return SPHX_OK;
    SPHX_CATCH
}

// wait for the stream, verify that every enqueued slab step really executed, report the clock
SPHX_EXPORT int sphx_slab_sync(sphx_ctx *c, sphx_status *status)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab, "SPHX:Slab:ctx", "not a slab context");
    SPHX_HIP(hipMemcpyAsync(c->h_clock, c->clock.get(), sizeof(Clock), hipMemcpyDeviceToHost, c->stream));
    SPHX_HIP(hipStreamSynchronize(c->stream));
    c->timer.collect();
    fill_status(c, status);
    const int64_t executed = (int64_t)c->h_clock->step - c->slab_step0;
    throw_on_status(c);
    if (executed != c->slab_steps_enqueued)
        throw Error(SPHX_ERR_STATE, "SPHX:Slab:overrun", "a slab step was enqueued after the loop had stopped");
    return SPHX_OK;
    SPHX_CATCH
}

// copy the current particles of the slab to host arrays of `capacity` entries; *n = count.  owned[i]
// tells whether particle i belongs to this rank (the rest are halo copies).
SPHX_EXPORT int sphx_slab_snapshot(sphx_ctx *c, int capacity, int *n, double *x, double *y, double *vx, double *vy,
                                   double *drho, int *id, int *owned)
{
    SPHX_TRY
    require(c != nullptr && c->is_slab && n != nullptr, "SPHX:Slab:ctx", "not a slab context");
    SPHX_HIP(hipMemcpyAsync(c->h_clock, c->clock.get(), sizeof(Clock), hipMemcpyDeviceToHost, c->stream));
    SPHX_HIP(hipStreamSynchronize(c->stream));
    const int m = c->h_clock->n;
    *n = m;
    require(capacity >= m, "SPHX:Slab:capacity", "snapshot arrays are too short");
    const bool skinned = c->rebuild_every > 1;
    const FluidSet fs = c->view(c->cur, skinned ? 0 : c->cur);
    hipStream_t s = c->stream;
    auto dl = [&](const void *src, void *dst, size_t bytes) {
        if (dst && m) SPHX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    };
    std::vector<double2> hp((size_t)std::max(m, 1)), hv((size_t)std::max(m, 1));
    std::vector<int> hcell((size_t)std::max(m, 1), 0);
    dl(fs.pos, hp.data(), (size_t)m * 16); dl(fs.vel, hv.data(), (size_t)m * 16);
    dl(fs.drho, drho, (size_t)m * 8); dl(fs.id, id, (size_t)m * 4);
    if (skinned) dl(fs.cell, hcell.data(), (size_t)m * 4);  // ownership goes by the binned column
    SPHX_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < m; ++i) {
        if (x) x[i] = hp[i].x;
        if (y) y[i] = hp[i].y;
        if (vx) vx[i] = hv[i].x;
        if (vy) vy[i] = hv[i].y;
        if (owned) {
            const int cx = hcell[i] / c->grid.ncy;
            owned[i] = skinned ? (cx >= c->grid.own_c0 && cx < c->grid.own_c1) : (hp[i].x >= c->grid.own_lo && hp[i].x < c->grid.own_hi);
        }
    }
    return SPHX_OK;
    SPHX_CATCH
}

// Average duration of one neighbour-pass kernel in the hipGraph-replay regime: `reps` back-to-back launches of
// that kernel alone are captured in a graph and replayed between two HIP events on the context's stream.  The
// passes only write per-step temporaries (continuity runs without its histogram), so the state is unchanged.
// This is the same quantity rocprofv3's kernel trace reports for the timed region (start of a dispatch to its
// end, dispatch gap included); the eager event-pair numbers of sphx_ctx_profile_* carry extra launch overhead
// at small particle counts.
SPHX_EXPORT int sphx_ctx_time_kernel(sphx_ctx *c, const char *name, int reps, double *avg_ms)
{
    SPHX_TRY
    require(c != nullptr && name != nullptr && avg_ms != nullptr && reps > 0, "SPHX:Ctx:null", "bad arguments");
    require(!c->is_slab, "SPHX:Ctx:slab", "not available on a slab context");
    const std::string n(name);
    int only = 0;
    if (n == "k_density" || n == "k_density_build" || n == "k_density_walk" || n == "k_density_dyn") only = 1;
    else if (n == "k_kgc") only = 2;
    else if (n == "k_forces") only = 3;
    else if (n == "k_continuity" || n == "k_continuity_clock") only = 4;
    else if (n == "k_continuity_density" && c->fuse_ea) only = 5;  // pass E and the next pass A in one launch
    require(only != 0, "SPHX:Ctx:kernel", "time_kernel knows k_density, k_kgc, k_forces, k_continuity (and k_continuity_density)");
    read_clock(c);
    if (c->h_clock->need_rebuild && c->h_clock->status == 0) forced_rebuild(c);
    const bool prof = c->profiling;
    c->profiling = false;
    hipGraph_t g = nullptr;
    hipGraphExec_t e = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    try {
        // arm run[cur] so the kernels execute; no step slot follows, so the clock does not advance
        arm_clock(c, c->prm.t_end, (long long)1, c->cur, (const double *)nullptr);
        const FluidSet fs = c->view(c->cur, c->lay);
        const int dmode = c->dyn ? 3 : (c->skin > 0.0 ? (c->pos == 0 ? 1 : 2) : 0);
        const FluidTmp &tt = c->tmp_par[c->fuse_ea ? c->cur : 0];  // (fuse_ea: the records / list of the current state parity)
        // make every temporary the timed kernel reads valid.  Where pass A of the coming step came with the last step's final
        // launch its list and records are there already, and stay: a timing call should not change what follows (the
        // stand-alone pass is a different kernel and may round differently in the last bit).
        if (c->fuse_ea && c->pos != 0) {
            for (int pass = 2; pass <= 4; ++pass) launch_physics_any(c, c->cur, fs, tt, 0, pass, dmode);
        } else {
            launch_physics_any(c, c->cur, fs, tt, 0, 0, dmode);
        }
        SPHX_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < reps; ++k) {
            if (only == 5) {  // (writes the other parity's list / records and the other state's drho: all rewritten by the next step)
                FluidTmp te = tt;
                const FluidSet o = c->view(1 - c->cur, c->lay);
                te.posn = o.pos; te.veln = o.vel; te.drhon = o.drho;
                launch_fused_ea(c, c->cur, fs, te, o, c->tmp_par[1 - c->cur], 0);
            } else {
                launch_physics_any(c, c->cur, fs, tt, 0, only, dmode);
            }
        }
        SPHX_HIP(hipStreamEndCapture(c->stream, &g));
        SPHX_HIP(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
        SPHX_HIP(hipGraphLaunch(e, c->stream));  // warm
        SPHX_HIP(hipEventCreate(&a));
        SPHX_HIP(hipEventCreate(&b));
        SPHX_HIP(hipEventRecord(a, c->stream));
        SPHX_HIP(hipGraphLaunch(e, c->stream));
        SPHX_HIP(hipEventRecord(b, c->stream));
        if (c->tail_clock)  // the timed passes stored plain maxima: back to "empty" for the next real step
            SPHX_HIP(hipMemsetAsync(c->vpart.get(), 0xFF, (size_t)c->n_vpart * sizeof(double), c->stream));
        SPHX_HIP(hipStreamSynchronize(c->stream));
        float ms = 0.f;
        SPHX_HIP(hipEventElapsedTime(&ms, a, b));
        *avg_ms = (double)ms / reps;
        // the timed passes overwrote the per-step outputs (rho, p, force, Vol, B) with values of a step that was never
        // taken: they are not downloadable until the next real step has produced them again
        c->have_step_outputs = false;
    } catch (...) {
        c->profiling = prof;
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        if (e) (void)hipGraphExecDestroy(e);
        if (g) (void)hipGraphDestroy(g);
        throw;
    }
    c->profiling = prof;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipGraphExecDestroy(e); (void)hipGraphDestroy(g);
    return SPHX_OK;
    SPHX_CATCH
}
