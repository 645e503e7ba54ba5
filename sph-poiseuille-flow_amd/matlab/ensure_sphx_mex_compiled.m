function ensure_sphx_mex_compiled(sphx_root, build_dir)
%ENSURE_SPHX_MEX_COMPILED  Drop-in for ensure_mex_compiled (SPH_Poiseuille.m S1): builds the three thin
% gateways against libsphx.so instead of compiling the reference's C sources.  Call once before S2:
%     ensure_sphx_mex_compiled('/path/to/this/repo', build_dir);  addpath(build_dir);
% after `python __graft_entry__.py` has produced sph-poiseuille-flow_amd/csrc/libsphx.so.
% The MEX names stay sph_neighbor_search_mex / sph_physics_shell_mex, so the rest of SPH_Poiseuille.m is untouched.
    inc = fullfile(sphx_root, 'include');
    libdir = fullfile(sphx_root, 'sph-poiseuille-flow_amd', 'csrc');
    src = fullfile(sphx_root, 'sph-poiseuille-flow_amd', 'matlab');
    if ~exist(fullfile(libdir, 'libsphx.so'), 'file')
        error('libsphx.so not found in %s: run python __graft_entry__.py first', libdir);
    end
    jobs = {'sph_neighbor_search_gateway.c', 'sph_neighbor_search_mex'; ...
            'sph_physics_shell_gateway.c',  'sph_physics_shell_mex'; ...
            'sphx_ctx_mex.c',               'sphx_ctx_mex'};
    for k = 1:size(jobs, 1)
        out_bin = fullfile(build_dir, [jobs{k, 2}, '.', mexext]);
        src_file = fullfile(src, jobs{k, 1});
        if exist(out_bin, 'file')
            a = dir(src_file); b = dir(out_bin);
            if a.datenum <= b.datenum, continue; end
        end
        fprintf('building %s\n', jobs{k, 2});
        mex('-R2018a', '-O', ['-I' inc], ['-L' libdir], '-lsphx', ['LDFLAGS=$LDFLAGS -Wl,-rpath,' libdir], ...
            '-output', jobs{k, 2}, '-outdir', build_dir, src_file);
    end
end
