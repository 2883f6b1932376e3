function [W, H, cost] = nmfx_nmfsc(V, num_basis_elems, config)
% nmfx_nmfsc  Drop-in for nmfsc(V, num_basis_elems, config) of the NMF Toolbox (Hoyer's NMF with sparseness constraints),
% computed on an AMD MI355X by libnmfx.  SOURCE ONLY (never run: no MATLAB in the build image; the gateway underneath is exercised
% by tests/test_mex_gateway.py).  Rename to nmfsc.m (ahead of the toolbox on the path) to drop in.
% The data check, the rescale V / max(V(:)), the initial projections onto the sparseness constraints and the line searches all run
% inside the library, in the order of the toolbox's nmfsc; this wrapper only supplies the random defaults (MATLAB's RNG stream
% stays MATLAB's) and prints what the toolbox prints.
if nargin < 3, config = struct; end
if min(V(:)) < 0, error('Negative values in data!'); end
[m, n] = size(V);
if ~isfield(config, 'W_init') || isempty(config.W_init), config.W_init = rand(m, num_basis_elems); end
if ~isfield(config, 'H_init') || isempty(config.H_init)
    config.H_init = rand(num_basis_elems, n);
    config.H_init = diag(1 ./ sqrt(sum(config.H_init.^2, 2))) * config.H_init;
end
if ~isfield(config, 'W_sparsity') || isempty(config.W_sparsity), config.W_sparsity = 0; end
if ~isfield(config, 'H_sparsity') || isempty(config.H_sparsity), config.H_sparsity = 0; end
if ~isfield(config, 'W_fixed') || isempty(config.W_fixed), config.W_fixed = false; end
if ~isfield(config, 'H_fixed') || isempty(config.H_fixed), config.H_fixed = false; end
if ~isfield(config, 'maxiter') || config.maxiter <= 0, config.maxiter = 100; end
if ~isfield(config, 'tolerance') || config.tolerance <= 0, config.tolerance = 1e-3; end
opts.sc_W_sparsity = double(config.W_sparsity);      % Hoyer sparseness targets in [0, 1] (values > 1 are clamped by the library like the toolbox)
opts.sc_H_sparsity = double(config.H_sparsity);
opts.W_fixed = uint8(logical(config.W_fixed));
opts.H_fixed = uint8(logical(config.H_fixed));
opts.maxiter = config.maxiter; opts.tolerance = config.tolerance;
if isfield(config, 'nmfx_device_ids'), opts.device_ids = int32(config.nmfx_device_ids); end   % extension: column shards over several GPUs
[W, H, cost, info] = nmfx_mex('nmfsc', double(V), double(config.W_init), double(config.H_init), int32(num_basis_elems), 1, opts);
if info.converged_early, display('Algorithm converged'); end   % the step size fell below 1e-200 in a line search
end
