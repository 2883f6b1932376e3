function [W, H, cost] = nmfx_lnmf(V, num_basis_elems, config)
% nmfx_lnmf  Drop-in for lnmf(V, num_basis_elems, config) of the NMF Toolbox (Local NMF, KL divergence), computed on an AMD MI355X by
% libnmfx.  SOURCE ONLY (never run: no MATLAB in the build image; the gateway underneath is exercised by tests/test_mex_gateway.py).
% Rename to lnmf.m (ahead of the toolbox on the path) to drop in.  One source, no sparsity options -- as in the toolbox.  Defaults
% follow the toolbox's local ValidateParameters: H_init = max(rand, eps), W_init = max(rand, eps) with unit-L1 columns.  The cost
% vector keeps maxiter entries (the toolbox's lnmf does not trim it on an early stop; the tail stays zero).
if nargin < 3, config = struct; end
[m, n] = size(V);
K = num_basis_elems;
if ~isfield(config, 'H_init') || isempty(config.H_init), config.H_init = max(rand(K, n), eps); end
if ~isfield(config, 'W_init') || isempty(config.W_init)
    w = max(rand(m, K), eps);
    config.W_init = w * diag(1 ./ sum(w, 1));
end
if ~isfield(config, 'W_fixed') || isempty(config.W_fixed), config.W_fixed = false; end
if ~isfield(config, 'H_fixed') || isempty(config.H_fixed), config.H_fixed = false; end
if ~isfield(config, 'maxiter') || config.maxiter <= 0, config.maxiter = 100; end
if ~isfield(config, 'tolerance') || config.tolerance <= 0, config.tolerance = 1e-3; end
opts.divergence = 1;                                   % KL: the only cost lnmf minimises
opts.W_fixed = uint8(logical(config.W_fixed)); opts.H_fixed = uint8(logical(config.H_fixed));
opts.maxiter = config.maxiter; opts.tolerance = config.tolerance;
if isfield(config, 'nmfx_device_ids'), opts.device_ids = int32(config.nmfx_device_ids); end   % extension: column shards over several GPUs
if isfield(config, 'nmfx_multi_backend'), opts.multi_backend = double(config.nmfx_multi_backend); end   % extension: 0 auto | 1 peer exchange | 2 RCCL all-reduce
[W, H, cost] = nmfx_mex('lnmf', double(V), double(config.W_init), double(config.H_init), int32(K), 1, opts);
end
