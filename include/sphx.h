/*
 * sphx.h -- C ABI of libsphx.so, the MI355X (gfx950) HIP implementation of the per-step SPH hot
 * path of KIYOYOZU/SPH-Poiseuille-Flow.
 *
 * Boundary being replaced: the two MEX gateways
 *     sph_neighbor_search_mex   (reference mex/sph_neighbor_search_mex.c:185, 5 in / 7 out)
 *     sph_physics_shell_mex     (reference mex/sph_physics_mex.c:1745, mode string + per-mode arity)
 * called from SPH_Poiseuille.m:167,169,366,380,388,398,406,419,428.
 *
 * Conventions (identical to the MEX surface):
 *   - every array is IEEE double, column-major: [n x 2] = x column then y column, B[n x 4] =
 *     B11|B12|B21|B22 columns (sph_physics_mex.c:362-365);
 *   - pair_i / pair_j hold 1-based particle indices stored as doubles
 *     (sph_neighbor_search_mex.c:375-376); fluid-fluid pairs appear once with i < j, fluid-wall
 *     pairs once with the fluid particle as i; wall particles are rows n_fluid..n_total-1;
 *   - inputs are borrowed and never written; outputs are caller-allocated with the sizes the MEX
 *     gateway would mxCreateDoubleMatrix.
 * All pointers are HOST pointers unless a name ends in _dev.  Functions return SPHX_OK (0) or a
 * negative status; sphx_last_error()/sphx_last_error_id() give the message and the MEX-style id
 * ("SPH:Neighbor:count", "SPH:Physics:int1:B" ...) a gateway passes to mexErrMsgIdAndTxt.
 * Thread model: like MATLAB, one caller thread per context; the library is not re-entrant on one
 * context.  There is no CPU fallback: without a HIP device every compute entry point fails with
 * SPHX_ERR_DEVICE.
 */
#ifndef SPHX_H
#define SPHX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHX_OK 0
#define SPHX_ERR_ARG (-1)      /* invalid argument (message carries the MEX id)                      */
#define SPHX_ERR_DEVICE (-2)   /* no HIP device / HIP runtime error                                  */
#define SPHX_ERR_STATE (-3)    /* call sequence error (e.g. fetch without search)                    */
#define SPHX_ERR_DIVERGED (-4) /* dt collapsed below 1e-14 (SPH_Poiseuille.m:260-263)                */
#define SPHX_ERR_GRID (-5)     /* particle left the cell grid / cell displacement bound violated      */

const char *sphx_version(void);
const char *sphx_last_error(void);
const char *sphx_last_error_id(void);
int sphx_device_count(void);
int sphx_set_device(int device);

/* ------------------------------------------------------------------------------------------------
 * 1. Stateless MEX-surface entry points (one per gateway call).  Host arrays in, host arrays out;
 *    the work runs in HIP kernels on the current device.
 * ---------------------------------------------------------------------------------------------- */

/* [pair_i,pair_j,dx,dy,r,W,dW] = sph_neighbor_search_mex(pos,n_fluid,n_total,h,DL)
 * (sph_neighbor_search_mex.c:12-28,185-421).  n_pairs is an output, so the call is two-phase: search
 * keeps the pair list in library-owned device memory and reports its length; fetch copies it into
 * caller arrays of at least that length and releases it.  Pairs come out ordered by i, then j. */
int sphx_neighbor_search(const double *pos, int n_fluid, int n_total, double h, double DL,
                         size_t *n_pairs);
int sphx_neighbor_fetch(double *pair_i, double *pair_j, double *dx, double *dy, double *r,
                        double *W, double *dW, size_t capacity);

/* 'density_correction' (sph_physics_mex.c:95-374): rho[n_total], Vol[n_total], B[n_total x 4]. */
int sphx_density_correction(size_t n_pairs, const double *pair_i, const double *pair_j,
                            const double *dx, const double *dy, const double *r, const double *W,
                            const double *dW, const double *mass, int n_fluid, int n_total,
                            double rho0, double h, double inv_sigma0, double *rho, double *Vol,
                            double *B);

/* 'viscous_force' (sph_physics_mex.c:396-550): force[n_total x 2]. */
int sphx_viscous_force(size_t n_pairs, const double *pair_i, const double *pair_j,
                       const double *dx, const double *dy, const double *r, const double *dW,
                       const double *vel, const double *Vol, const double *B, double mu, double h,
                       int n_fluid, int n_total, const double *mass, const double *wall_vel,
                       double *force);

/* 'transport_correction' (sph_physics_mex.c:569-714): pos_out[n_total x 2].  The 13-argument MEX
 * form uses transport_coeff = 0.2 (:584); the gateway passes that default explicitly. */
int sphx_transport_correction(size_t n_pairs, const double *pair_i, const double *pair_j,
                              const double *dx, const double *dy, const double *r,
                              const double *dW, const double *Vol, const double *B,
                              const double *pos, double h, int n_fluid, int n_total,
                              double transport_coeff, double *pos_out);

/* 'integration_1st' (sph_physics_mex.c:736-967): rho, p, pos, force, drho(diss). */
int sphx_integration_1st(size_t n_pairs, const double *pair_i, const double *pair_j,
                         const double *dx, const double *dy, const double *r, const double *dW,
                         const double *Vol, const double *B, const double *rho, const double *mass,
                         const double *pos, const double *vel, const double *drho_dt,
                         const double *force_prior, double dt, int n_fluid, int n_total,
                         double rho0, double p0, double c_f, const double *wall_vel,
                         double *rho_out, double *p_out, double *pos_out, double *force_out,
                         double *drho_out);

/* 'integration_2nd' (sph_physics_mex.c:987-1119): pos, drho_dt, zeros[n_total x 2] (may be NULL). */
int sphx_integration_2nd(size_t n_pairs, const double *pair_i, const double *pair_j,
                         const double *dx, const double *dy, const double *r, const double *dW,
                         const double *Vol, const double *rho, const double *pos,
                         const double *vel, double dt, int n_fluid, int n_total,
                         const double *wall_vel, double *pos_out, double *drho_out,
                         double *zeros_out);

/* 'integration_verlet' (sph_physics_mex.c:1316-1469): rho, p, pos, vel, drho_dt, force. */
int sphx_integration_verlet(size_t n_pairs, const double *pair_i, const double *pair_j,
                            const double *dx, const double *dy, const double *r, const double *dW,
                            const double *Vol, const double *B, const double *rho,
                            const double *mass, const double *pos, const double *vel,
                            const double *drho_dt, const double *force_prior, double dt,
                            int n_fluid, int n_total, double rho0, double p0, double c_f,
                            const double *wall_vel, double *rho_out, double *p_out,
                            double *pos_out, double *vel_out, double *drho_out, double *force_out);

/* 'advance_shell_step' (sph_physics_mex.c:1490-1639): density -> viscous(+mass*g) -> transport(0.2)
 * -> verlet in one call; 9 outputs rho,p,pos,vel,drho_dt,force,force_prior,Vol,B (:1623-1631). */
int sphx_advance_shell_step(size_t n_pairs, const double *pair_i, const double *pair_j,
                            const double *dx, const double *dy, const double *r, const double *W,
                            const double *dW, const double *mass, const double *pos,
                            const double *vel, const double *wall_vel, const double *rho,
                            const double *drho_dt, double dt, int n_fluid, int n_total,
                            double rho0, double p0, double c_f, double mu, double h,
                            double inv_sigma0, double gravity_g, double *rho_out, double *p_out,
                            double *pos_out, double *vel_out, double *drho_out, double *force_out,
                            double *force_prior_out, double *Vol_out, double *B_out);

/* 'wall_shear_monitor' (sph_physics_mex.c:1653-1743): tau_bottom, tau_top. */
int sphx_wall_shear_monitor(size_t n_pairs, const double *pair_i, const double *pair_j,
                            const double *dx, const double *dy, const double *r, const double *dW,
                            const double *pos, const double *vel, const double *wall_vel,
                            const double *Vol, const double *B, int n_fluid, int n_total, double DL,
                            double DH, double mu, double h, double *tau_bottom, double *tau_top);

/* ------------------------------------------------------------------------------------------------
 * 2. Device-resident context: the whole step loop of SPH_Poiseuille.m:250-292 stays in HBM.
 *    One context = one x-slab of the channel on one GPU (slab == whole channel for one GPU).
 * ---------------------------------------------------------------------------------------------- */

typedef struct sphx_ctx sphx_ctx;

typedef struct sphx_params {
    /* physical / numerical constants, SPH_Poiseuille.m:46-80 */
    double DL, DH, dp, h;
    double rho0, mu, c_f, p0, inv_sigma0, gravity_g;
    double transport_coeff; /* 0.30 in the main loop (:77); 0.2 reproduces advance_shell_step        */
    double t_end;           /* loop bound, SPH_Poiseuille.m:247                                     */
    int32_t sort_interval;  /* kept for signature parity; the device keeps cell order (see rebuild_every) */
    int32_t lanes_per_particle; /* 0 = auto; 1,2,4,8,16,32: lanes cooperating on one neighbour ring */
    int32_t steps_per_graph;    /* 0 = auto; steps captured per hipGraph replay (even)               */
    int32_t dual_rate;          /* 0 / 1 = the reference's single-rate loop (SPH_Poiseuille.m:250-292, the parity path).
                                   2..4 = opt-in dual-rate loop for small channels (<= ~30 k fluid particles, 16 / 32
                                   lanes per particle): one step slot is an OUTER step -- density summation, KGC, viscous
                                   force, transport shift once -- of up to dual_rate acoustic sub-steps of pressure /
                                   continuity.  The count is fixed per context: as many acoustic steps as fit into the
                                   viscous / body-force step (1 on fine channels, which are viscous-limited, and on
                                   contexts that are not eligible; sphx_ctx_substeps reports it).  Not reference
                                   behaviour: validated against the analytic profile only (2 sub-steps reproduce the
                                   single-rate L2 and wall shear at dp = 0.05 / 0.025 / 0.02; 4 is noisier at dp = 0.05). */
    int32_t rebuild_every;      /* 0 = auto; K >= 1: particles are re-binned into cells every K-th step; in
                                   between, sweeps are centred on the cell a particle was binned into and
                                   the cells carry a skin (results do not depend on K beyond summation
                                   order: the device stops and re-bins before any neighbour can be missed) */
    int32_t dynamic_rebin;      /* device-decided re-binning (no host round trips): 0 = by size (on from 2 x 10^6 fluid
                                   particles), 1 = on, 2 = off                                              */
    double skin_h;              /* cell skin in units of h for K > 1; <= 0 = sized from K.  Both left at 0: the
                                   measured pair for the size class (K = 16 up to 20 k fluid particles, 8 up to 300 k,
                                   10 with 0.42 h up to 2 x 10^6, 24 with 0.28 h where the device re-bins by itself)  */
} sphx_params;

typedef struct sphx_status {
    double t;            /* simulated time reached                                                  */
    double dt_last;      /* dt of the last completed step                                           */
    double dt_next;      /* dt the next step would use                                              */
    double vmax;         /* max |v| over fluid after the last step (SPH_Poiseuille.m:286)           */
    int64_t step;        /* steps completed since creation (state.step)                             */
    int32_t done;        /* 1 when t >= t_target - 1e-12                                            */
    int32_t device_status; /* 0 ok, else SPHX_ERR_DIVERGED / SPHX_ERR_GRID raised on device          */
} sphx_status;

/* Create a context from host state in MEX layout.  pos/vel/wall_vel [n_total x 2]; drho_dt, mass
 * [n_total].  Rows 0..n_fluid-1 fluid, the rest wall (SPH_Poiseuille.m:107).  Builds the cell grid
 * (the neighbour structure of SPH_Poiseuille.m:167) on the device. */
int sphx_ctx_create(sphx_ctx **ctx, const sphx_params *prm, int n_fluid, int n_total,
                    const double *pos, const double *vel, const double *drho_dt,
                    const double *mass, const double *wall_vel, double t0, int64_t step0);
void sphx_ctx_destroy(sphx_ctx *ctx);

/* Run steps until t >= t_target - 1e-12 (one pass of the inner while of SPH_Poiseuille.m:250; dt is
 * clipped by remain = min(t_target - t, t_end - t), :252) or until max_steps steps have been taken
 * (max_steps <= 0: unlimited).  Blocks until the device is idle. */
int sphx_ctx_advance(sphx_ctx *ctx, double t_target, int64_t max_steps, sphx_status *status);

/* Enqueue exactly n_steps steps without host synchronisation (benchmark / pipelined use), dt clipped
 * only by t_end.  sphx_ctx_sync waits and reports. */
int sphx_ctx_enqueue_steps(sphx_ctx *ctx, int64_t n_steps);
int sphx_ctx_sync(sphx_ctx *ctx, sphx_status *status);

/* Steps are replayed as hipGraphs captured per (schedule phase, batch length); a combination that has not come up
 * before is captured on first use (a few ms).  sphx_ctx_prepare_steps captures, without running anything, what
 * an sphx_ctx_enqueue_steps(n_steps) / sphx_ctx_advance(.., max_steps = n_steps) issued next would replay, so
 * that the call itself is pure replay.  sphx_ctx_graph_stats: step slots replayed from graphs / launched
 * eagerly / graphs captured since creation. */
int sphx_ctx_prepare_steps(sphx_ctx *ctx, int64_t n_steps);
int sphx_ctx_graph_stats(sphx_ctx *ctx, int64_t *slots_replayed, int64_t *slots_eager,
                         int64_t *graphs_captured);

/* Copy state back in the caller's original row order.  Any pointer may be NULL.  rho,p,force,
 * force_prior,Vol,B are those of the last completed step (what integration_verlet / density_correction
 * returned in SPH_Poiseuille.m:254-266).  Like sphx_ctx_sync, download and monitor first wait for everything
 * enqueued and take the steps a batch of sphx_ctx_enqueue_steps still owes (a batch stops early when the cell
 * grid goes stale): "last completed step" is the last step of everything asked for, and the state and the
 * step outputs handed out belong to that same step. */
int sphx_ctx_download(sphx_ctx *ctx, double *pos, double *vel, double *rho, double *p,
                      double *drho_dt, double *force, double *force_prior, double *Vol, double *B);

/* Monitors of SPH_Poiseuille.m:281-291 on the current neighbour structure: wall shear (new pairs,
 * new pos/vel, previous Vol/B), pair count of the current list. */
int sphx_ctx_monitor(sphx_ctx *ctx, double *tau_bottom, double *tau_top, double *n_pairs);

/* The neighbour list the context currently holds, in the MEX convention and the caller's original
 * row numbering (two-phase like sphx_neighbor_search / sphx_neighbor_fetch). */
int sphx_ctx_neighbor_list(sphx_ctx *ctx, size_t *n_pairs);

/* Average device time (ms) of each per-step kernel since the last call, measured with HIP events on
 * the context's stream when profiling is enabled.  names: array of const char* filled by the library. */
int sphx_ctx_profile_enable(sphx_ctx *ctx, int on);
int sphx_ctx_profile_read(sphx_ctx *ctx, int capacity, const char **names, double *avg_ms,
                          int64_t *launches, int *n_kernels);

/* Average duration (ms) of ONE neighbour-pass kernel ("k_density", "k_kgc", "k_forces", "k_continuity") in
 * the hipGraph-replay regime: `reps` back-to-back launches between two HIP events.  pos/vel/drho_dt are left
 * unchanged; the per-step outputs (rho, p, force, Vol, B) are overwritten and cannot be downloaded again until
 * the next step has been taken. */
int sphx_ctx_time_kernel(sphx_ctx *ctx, const char *name, int reps, double *avg_ms);

/* Cell-grid policy in force: re-binning interval K (constant), skin in length units, the number of unscheduled
 * re-binnings so far (the drift bound was hit: host-driven contexts then re-bin every step for a cool-down of
 * 16..1024 steps, dynamic contexts just re-bin), and the largest distance of any particle from where it was
 * binned (as of the last advance / sync; always <= skin/2). */
int sphx_ctx_grid_policy(sphx_ctx *ctx, int *rebuild_every, double *skin, int64_t *forced_rebuilds,
                         double *drift);

/* The launch shape the context chose (lanes cooperating per particle, steps per hipGraph replay). */
int sphx_ctx_tuning(sphx_ctx *ctx, int *lanes_per_particle, int *steps_per_graph);

/* The launch schedule the context runs: fuse_ea = pass E of a step and pass A of the next one share a launch
 * (three launches per step), tail_clock = the clock update rides in the last launch of a step, dynamic = the device
 * decides when to re-bin; rebins = re-binnings carried out by step slots so far (scheduled ones; dynamic contexts:
 * all of them), not counting the forced ones sphx_ctx_grid_policy reports. */
int sphx_ctx_schedule(sphx_ctx *ctx, int *fuse_ea, int *tail_clock, int *dynamic, int64_t *rebins);

/* The kernel forms the context runs: walk_kernels = the large-channel ("_w") passes with 16-bit neighbour lists (up to 8
 * lanes per particle), lds_tiles = the force pass stages its workgroup's neighbourhood in LDS, tiles_abe = passes A, B and E
 * do too, coded_lists = the lists name tile slots instead of index differences. */
int sphx_ctx_kernel_forms(sphx_ctx *ctx, int *walk_kernels, int *lds_tiles, int *tiles_abe, int *coded_lists);

/* Inner sub-steps per step slot: 1 unless sphx_params::dual_rate asked for the dual-rate loop and the context is
 * eligible.  With n_inner > 1 a "step" of sphx_status / max_steps is an outer step (t advances by n_inner * dt_last). */
int sphx_ctx_substeps(sphx_ctx *ctx, int *n_inner);

/* Global particle counts and the cell grid the context built. */
int sphx_ctx_info(sphx_ctx *ctx, int *n_fluid, int *n_wall, int *n_cell_x, int *n_cell_y);

/* ------------------------------------------------------------------------------------------------
 * 3. x-slab contexts (multi-GPU).  The channel is cut into n_ranks slabs of whole cell columns; each
 *    rank (one process per GPU) holds its columns plus halo_cols columns of copies on either side.
 *    The reference has no counterpart (single process).  One step is
 *        sphx_slab_compute -> {exchange two messages with the ring neighbours, all-reduce max|v|}
 *        -> sphx_slab_finish
 *    either driven by the caller with its own transport (the three calls below) or by the library's native loop
 *    over RCCL (sphx_slab_run).
 *    All *_dev pointers are DEVICE pointers (e.g. torch tensors) and every call is asynchronous on the
 *    context's stream (hip_stream of sphx_slab_create, or an internal one when NULL).
 * ---------------------------------------------------------------------------------------------- */

/* Create rank `rank` of `n_ranks` from the GLOBAL host state (same arguments as sphx_ctx_create on
 * every rank).  halo_cols >= 4.  prm->rebuild_every selects the protocol: 1 = the slab re-bins every step and is
 * driven by the caller (sphx_slab_local_vmax / _prepare / _compute / _finish below, any transport); any other value
 * (0 = auto: 5) = skinned slab for the library's own loops only (sphx_slab_run over RCCL, sphx_slab_group_run): it
 * re-bins every K-th step or when the device sees the drift bound hit, and the four caller-driven calls refuse it
 * with SPHX:Slab:protocol. */
int sphx_slab_create(sphx_ctx **ctx, const sphx_params *prm, int n_fluid, int n_total,
                     const double *pos, const double *vel, const double *drho_dt, const double *mass,
                     const double *wall_vel, double t0, int64_t step0, int rank, int n_ranks,
                     int halo_cols, void *hip_stream);
/* Message length in doubles (1 + 7*capacity: count, then x,y,vx,vy,drho,mass,id blocks), the owned
 * global cell columns [col0,col1), current local particle count and array capacity. */
int sphx_slab_layout(sphx_ctx *ctx, int64_t *msg_doubles, int *col0, int *col1, int *n_local,
                     int *capacity);
/* max |v| over the owned particles of the current state -> vmax_dev[0]. */
int sphx_slab_local_vmax(sphx_ctx *ctx, double *vmax_dev);
/* Arm the device clock for a run to t_target / at most max_steps (<=0: unlimited) steps; the first dt
 * uses vmax_global_dev[0] (the all-reduced value). */
int sphx_slab_prepare(sphx_ctx *ctx, double t_target, int64_t max_steps, const double *vmax_global_dev);
/* First half of a step: the four neighbour passes, local max |v| -> vmax_local_dev[0], and the two
 * outgoing messages (left / right ring neighbour). */
int sphx_slab_compute(sphx_ctx *ctx, double *send_left_dev, double *send_right_dev,
                      double *vmax_local_dev);
/* Second half: take the messages received from the left / right neighbour and the global max |v|,
 * update the clock and rebuild the cell grid. */
int sphx_slab_finish(sphx_ctx *ctx, const double *recv_left_dev, const double *recv_right_dev,
                     const double *vmax_global_dev);
/* Native step loop.  sphx_slab_run enqueues n_steps whole steps (compute -> two sends + two receives with the ring
 * neighbours and an 8-byte max all-reduce through RCCL -> finish) on the context's stream, in library-owned message
 * buffers; nothing of the host language runs between steps.  One context per process (one process per GPU): rank 0
 * makes an id with sphx_comm_unique_id (128 bytes), the launcher carries it to the other ranks (torch.distributed
 * broadcast, MPI, a file ...), every rank joins with sphx_slab_comm_init.  librccl is loaded at that moment.
 * sphx_slab_group_run: all slabs of the ring in ONE process on one device -- the same loop with device-to-device
 * copies as the transport (tests and rehearsals on a one-GPU box).  Both return without waiting; sphx_slab_sync
 * waits and reports. */
/* SPHX_OK when librccl can be loaded with every entry point the native loop needs (purely local, no communication):
 * a launcher lets all ranks agree on this BEFORE anyone enters the collective sphx_slab_comm_init. */
int sphx_comm_available(void);
int sphx_comm_unique_id(void *id_bytes, int capacity);
/* Diagnostic: runs the exchange pattern of sphx_slab_run (grouped sends / receives, the all-reduce) on a one-rank
 * communicator on the current device and checks what comes back.  SPHX_OK, or an error naming the RCCL call that failed. */
int sphx_comm_selftest(void);
/* The same calls captured into a hipGraph and replayed (one-rank communicator): can this RCCL be captured?  What
 * sphx_slab_graph_prepare relies on. */
int sphx_comm_selftest_graph(void);
int sphx_slab_comm_init(sphx_ctx *ctx, const void *id_bytes);
int sphx_slab_comm_destroy(sphx_ctx *ctx);
int sphx_slab_run(sphx_ctx *ctx, double t_target, int64_t n_steps);
int sphx_slab_group_run(sphx_ctx **ctxs, int n_ranks, double t_target, int64_t n_steps);
/* Capture ten whole steps of the native loop -- kernels and the RCCL calls (n_ranks = 1: the context of this process's
 * rank) or the copies and cross-stream dependencies of an in-process ring (n_ranks >= 2) -- into one hipGraph; the two
 * loops above then replay it for every full batch of ten steps.  Call after the loop has run at least two steps, on
 * every rank at the same point (the graph is warmed with one idle replay, which communicates).  Skinned slabs only. */
int sphx_slab_graph_prepare(sphx_ctx **ctxs, int n_ranks);
/* Wait for the stream; fails if a step was enqueued after the loop had stopped or a buffer overflowed. */
int sphx_slab_sync(sphx_ctx *ctx, sphx_status *status);
/* Host copy of the slab's current particles (owned + halo copies); owned[i] = 1 for owned ones. */
int sphx_slab_snapshot(sphx_ctx *ctx, int capacity, int *n, double *x, double *y, double *vx,
                       double *vy, double *drho, int *id, int *owned);

#ifdef __cplusplus
}
#endif
#endif /* SPHX_H */
