/*
 * sphx_ctx_mex.c -- optional third gateway: the device-resident loop (sphx_ctx_* of include/sphx.h) for MATLAB.
 *   h  = sphx_ctx_mex('create', cfg, n_fluid, n_total, pos, vel, drho_dt, mass, wall_vel, t, step)
 *   st = sphx_ctx_mex('advance', h, t_target, max_steps)        % struct: t, dt_last, dt_next, vmax, step, done
 *   [pos,vel,rho,p,drho_dt,force,force_prior,Vol,B] = sphx_ctx_mex('download', h)
 *   [tau_bottom, tau_top, n_pairs] = sphx_ctx_mex('monitor', h)
 *   sphx_ctx_mex('prepare', h, n_steps)      % capture the graph an advance(h, t, n_steps) issued next replays
 *   [replayed, eager, captured] = sphx_ctx_mex('graph_stats', h)
 *   sphx_ctx_mex('destroy', h)
 * cfg is the struct SPH_Poiseuille.m builds at :175-196 (fields DL, DH, dp, h, rho0, mu, c_f, p0, inv_sigma0,
 * gravity_g, transport_coeff, t_end, sort_interval).  Never built with MATLAB in this repository (there is none in the
 * image): tests/test_matlab_gateways.py compiles it against a mock of the C Matrix/MEX API (tests/stubs) and checks it
 * against sph-poiseuille-flow_amd/driver.py (engine="resident"), which is the same loop.
 */
#include <string.h>
#include "mex.h"
#include "sphx.h"

static void ok(int rc) { if (rc != SPHX_OK) mexErrMsgIdAndTxt(sphx_last_error_id(), "%s", sphx_last_error()); }
static void arity(const char *cmd, int nrhs, int want_rhs, int nlhs, int max_lhs)
{
    if (nrhs != want_rhs) mexErrMsgIdAndTxt("SPHX:Ctx:nrhs", "%s expects %d inputs after the command, got %d", cmd, want_rhs - 1, nrhs - 1);
    if (nlhs > max_lhs) mexErrMsgIdAndTxt("SPHX:Ctx:nlhs", "%s returns at most %d outputs", cmd, max_lhs);
}
static double fld(const mxArray *s, const char *name)
{
    const mxArray *f = mxGetField(s, 0, name);
    if (!f) mexErrMsgIdAndTxt("SPHX:Ctx:cfg", "cfg is missing field %s", name);
    return mxGetScalar(f);
}
static sphx_ctx *handle(const mxArray *a)
{
    if (mxIsDouble(a) || mxIsChar(a) || mxGetNumberOfElements(a) != 1)
        mexErrMsgIdAndTxt("SPHX:Ctx:handle", "second argument must be the uint64 handle 'create' returned");
    return (sphx_ctx *)(uintptr_t)(*(uint64_t *)mxGetData(a));
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    char cmd[32];
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("SPHX:Ctx:cmd", "first argument must be a command string");
    mxGetString(prhs[0], cmd, sizeof(cmd));
    if (strcmp(cmd, "create") == 0) {
        sphx_params p;
        sphx_ctx *c = NULL;
        const mxArray *cfg;
        int k;
        arity(cmd, nrhs, 11, nlhs, 1);
        cfg = prhs[1];
        for (k = 4; k <= 8; ++k)
            if (!mxIsDouble(prhs[k])) mexErrMsgIdAndTxt("SPHX:Ctx:type", "create: argument %d must be a double array", k + 1);
        memset(&p, 0, sizeof(p));
        p.DL = fld(cfg, "DL"); p.DH = fld(cfg, "DH"); p.dp = fld(cfg, "dp"); p.h = fld(cfg, "h"); p.rho0 = fld(cfg, "rho0");
        p.mu = fld(cfg, "mu"); p.c_f = fld(cfg, "c_f"); p.p0 = fld(cfg, "p0"); p.inv_sigma0 = fld(cfg, "inv_sigma0");
        p.gravity_g = fld(cfg, "gravity_g"); p.transport_coeff = fld(cfg, "transport_coeff"); p.t_end = fld(cfg, "t_end");
        p.sort_interval = (int32_t)fld(cfg, "sort_interval");
        if (mxGetField(cfg, 0, "dual_rate")) p.dual_rate = (int32_t)fld(cfg, "dual_rate");  /* optional, see sphx.h */
        ok(sphx_ctx_create(&c, &p, (int)mxGetScalar(prhs[2]), (int)mxGetScalar(prhs[3]), mxGetDoubles(prhs[4]),
                           mxGetDoubles(prhs[5]), mxGetDoubles(prhs[6]), mxGetDoubles(prhs[7]), mxGetDoubles(prhs[8]),
                           mxGetScalar(prhs[9]), (int64_t)mxGetScalar(prhs[10])));
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)c;
        mexLock();
    } else if (strcmp(cmd, "advance") == 0) {
        static const char *names[] = {"t", "dt_last", "dt_next", "vmax", "step", "done"};
        sphx_status st;
        arity(cmd, nrhs, 4, nlhs, 1);
        ok(sphx_ctx_advance(handle(prhs[1]), mxGetScalar(prhs[2]), (int64_t)mxGetScalar(prhs[3]), &st));
        plhs[0] = mxCreateStructMatrix(1, 1, 6, names);
        mxSetField(plhs[0], 0, "t", mxCreateDoubleScalar(st.t));
        mxSetField(plhs[0], 0, "dt_last", mxCreateDoubleScalar(st.dt_last));
        mxSetField(plhs[0], 0, "dt_next", mxCreateDoubleScalar(st.dt_next));
        mxSetField(plhs[0], 0, "vmax", mxCreateDoubleScalar(st.vmax));
        mxSetField(plhs[0], 0, "step", mxCreateDoubleScalar((double)st.step));
        mxSetField(plhs[0], 0, "done", mxCreateDoubleScalar((double)st.done));
    } else if (strcmp(cmd, "download") == 0) {
        int nf = 0, nw = 0, nt, k;
        static const int cols[9] = {2, 2, 1, 1, 1, 2, 2, 1, 4};
        double *out[9] = {0};
        arity(cmd, nrhs, 2, nlhs, 9);
        ok(sphx_ctx_info(handle(prhs[1]), &nf, &nw, NULL, NULL));
        nt = nf + nw;
        for (k = 0; k < 9 && k < (nlhs > 0 ? nlhs : 1); ++k) {
            plhs[k] = mxCreateDoubleMatrix((mwSize)nt, (mwSize)cols[k], mxREAL);
            out[k] = mxGetDoubles(plhs[k]);
        }
        ok(sphx_ctx_download(handle(prhs[1]), out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], out[8]));
    } else if (strcmp(cmd, "monitor") == 0) {
        double tb = 0.0, tt = 0.0, np = 0.0;
        arity(cmd, nrhs, 2, nlhs, 3);
        ok(sphx_ctx_monitor(handle(prhs[1]), &tb, &tt, nlhs > 2 ? &np : NULL));
        plhs[0] = mxCreateDoubleScalar(tb);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(tt);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(np);
    } else if (strcmp(cmd, "prepare") == 0) {
        arity(cmd, nrhs, 3, nlhs, 0);
        ok(sphx_ctx_prepare_steps(handle(prhs[1]), (int64_t)mxGetScalar(prhs[2])));
    } else if (strcmp(cmd, "graph_stats") == 0) {
        int64_t a = 0, b = 0, g = 0;
        arity(cmd, nrhs, 2, nlhs, 3);
        ok(sphx_ctx_graph_stats(handle(prhs[1]), &a, &b, &g));
        plhs[0] = mxCreateDoubleScalar((double)a);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)b);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar((double)g);
    } else if (strcmp(cmd, "destroy") == 0) {
        arity(cmd, nrhs, 2, nlhs, 0);
        sphx_ctx_destroy(handle(prhs[1]));
        mexUnlock();
    } else {
        mexErrMsgIdAndTxt("SPHX:Ctx:cmd", "unknown command %s", cmd);
    }
}
