function [W, H, cost] = nmfx_cnmfsc(V, num_basis_elems, context_len, config)
% nmfx_cnmfsc  Drop-in for cnmfsc(V, num_basis_elems, context_len, config) of the NMF Toolbox (convolutive NMF with Hoyer sparseness
% constraints), computed on an AMD MI355X by libnmfx.  SOURCE ONLY (never run: no MATLAB in the build image; the gateway underneath
% is exercised by tests/test_mex_gateway.py).  Rename to cnmfsc.m (ahead of the toolbox on the path) to drop in.
% The rescale V / max(V(:)), the initial projections and the line searches run inside the library in the toolbox's order, including
% its quirks (W_init is not projected at the start; the sparse-W line search ends by step-size underflow); this wrapper supplies the
% random defaults so that MATLAB's RNG stream stays MATLAB's.
if nargin < 4, config = struct; end
if min(V(:)) < 0, error('Negative values in data!'); end
[m, n] = size(V);
K = num_basis_elems; T = context_len;
if ~isfield(config, 'W_init') || isempty(config.W_init), config.W_init = rand(m, K, T); end
if ~isfield(config, 'H_init') || isempty(config.H_init)
    h = rand(K, n);
    config.H_init = diag(1 ./ sqrt(sum(h.^2, 2))) * h;
end
if ~isfield(config, 'W_sparsity') || isempty(config.W_sparsity), config.W_sparsity = 0; end
if ~isfield(config, 'H_sparsity') || isempty(config.H_sparsity), config.H_sparsity = 0; end
if ~isfield(config, 'W_fixed') || isempty(config.W_fixed), config.W_fixed = false; end
if ~isfield(config, 'H_fixed') || isempty(config.H_fixed), config.H_fixed = false; end
if ~isfield(config, 'maxiter') || config.maxiter <= 0, config.maxiter = 100; end
if ~isfield(config, 'tolerance') || config.tolerance <= 0, config.tolerance = 1e-3; end
opts.sc_W_sparsity = double(config.W_sparsity);      % Hoyer sparseness targets in [0, 1]
opts.sc_H_sparsity = double(config.H_sparsity);
opts.W_fixed = uint8(logical(config.W_fixed)); opts.H_fixed = uint8(logical(config.H_fixed));
opts.maxiter = config.maxiter; opts.tolerance = config.tolerance;
[W, H, cost, info] = nmfx_mex('cnmfsc', double(V), double(config.W_init), double(config.H_init), int32(K), T, opts);
if info.converged_early, display('Algorithm converged'); end
end
