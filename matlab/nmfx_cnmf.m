function [W, H, cost] = nmfx_cnmf(V, num_basis_elems, context_len, config)
% nmfx_cnmf  Drop-in for cnmf(V, num_basis_elems, context_len, config) of the NMF Toolbox, computed on an AMD MI355X by libnmfx.
% SOURCE ONLY (never run: no MATLAB in the build image; the gateway underneath is exercised by tests/test_mex_gateway.py).
% Rename to cnmf.m (ahead of the toolbox on the path) to drop in.  Argument meaning, defaults, cell handling and error messages
% follow the toolbox's cnmf and its local ValidateParameters; the library does the normalisation of the init and the iterations.
if nargin < 4, config = struct; end
if ~iscell(num_basis_elems), num_basis_elems = {num_basis_elems}; end
S = numel(num_basis_elems);
T = context_len;
[m, n] = size(V);
% --- defaults exactly as the toolbox's local ValidateParameters ---
if ~isfield(config, 'divergence'), config.divergence = 'euclidean'; end
is_ab = any(strcmp(config.divergence, {'ab_divergence', 'ab'}));
if ~isfield(config, 'alpha') || ~is_ab, config.alpha = 1; end
if ~isfield(config, 'beta') || ~is_ab, config.beta = 1; end
if is_ab && config.alpha == 0 && config.beta == 0, error('alpha = 0 and beta = 0 is not supported at this time.'); end
switch config.divergence                     % the toolbox's cnmf maps the name to (alpha, beta) and has NO otherwise branch:
    case 'euclidean', dv = 0;                % any other string runs the euclidean updates with an all-zero cost vector
    case {'kl_divergence', 'kl'}, dv = 1;
    case {'is_divergence', 'is'}, dv = 2;
    case {'ab_divergence', 'ab'}, dv = 3;
    otherwise, dv = 4;                       % 'frobenius' and unknown names: NMFX_DIV_EUCLIDEAN_NOCOST
end
if ~isfield(config, 'H_init') || isempty(config.H_init)
    is_H_cell = S > 1; config.H_init = cell(S, 1);
    for s = 1 : S, config.H_init{s} = max(rand(num_basis_elems{s}, n), eps); end
elseif iscell(config.H_init) && numel(config.H_init) ~= S
    error(['Requested ', num2str(S), ' sources. Given ', num2str(numel(config.H_init)), ' initial encoding matrices.']);
elseif ~iscell(config.H_init), is_H_cell = false; config.H_init = {config.H_init};
else, is_H_cell = true; config.H_init = config.H_init(:); end
if ~isfield(config, 'W_init') || isempty(config.W_init)
    is_W_cell = S > 1; config.W_init = cell(1, S);
    for s = 1 : S
        w = rand(m, num_basis_elems{s}, T);
        for k = 1 : num_basis_elems{s}
            w(:, k, :) = w(:, k, :) / (norm(squeeze(w(:, k, :)), 'fro') / T);
        end
        config.W_init{s} = w;
    end
elseif iscell(config.W_init) && numel(config.W_init) ~= S
    error(['Requested ', num2str(S), ' sources. Given ', num2str(numel(config.W_init)), ' initial basis matrices.']);
elseif ~iscell(config.W_init), is_W_cell = false; config.W_init = {config.W_init};
else, is_W_cell = true; config.W_init = config.W_init(:)'; end
opts.W_sparsity = nmfx_per_source(config, 'W_sparsity', S, 0, true, 'sparsity levels');
opts.H_sparsity = nmfx_per_source(config, 'H_sparsity', S, 0, true, 'sparsity levels');
opts.W_fixed = uint8(nmfx_per_source(config, 'W_fixed', S, 0, false, 'update switches'));
opts.H_fixed = uint8(nmfx_per_source(config, 'H_fixed', S, 0, false, 'update switches'));
if ~isfield(config, 'maxiter') || config.maxiter <= 0, config.maxiter = 100; end
if ~isfield(config, 'tolerance') || config.tolerance <= 0, config.tolerance = 1e-3; end
opts.divergence = dv; opts.alpha = config.alpha; opts.beta = config.beta;
opts.maxiter = config.maxiter; opts.tolerance = config.tolerance;
if isfield(config, 'nmfx_device_ids'), opts.device_ids = int32(config.nmfx_device_ids); end   % extension: column shards over several GPUs
if isfield(config, 'nmfx_multi_backend'), opts.multi_backend = double(config.nmfx_multi_backend); end   % extension: 0 auto | 1 peer exchange | 2 RCCL all-reduce
K_s = int32(cell2mat(num_basis_elems(:)'));
W_all = cat(2, config.W_init{:});            % cell2mat(1 x S) of m x K_s x T tensors: along dimension 2
[Wa, Ha, cost] = nmfx_mex('cnmf', double(V), double(W_all), double(cell2mat(config.H_init)), K_s, T, opts);
edges = [0, cumsum(double(K_s))];
W = cell(1, S); H = cell(S, 1);
for s = 1 : S
    W{s} = Wa(:, edges(s)+1 : edges(s+1), :);
    H{s} = Ha(edges(s)+1 : edges(s+1), :);
end
if ~is_W_cell, W = W{1}; end
if ~is_H_cell, H = H{1}; end
end

function v = nmfx_per_source(config, name, S, dflt, clamp, what)
if ~isfield(config, name) || isempty(config.(name)), v = repmat(dflt, 1, S); return; end
x = config.(name);
if iscell(x) && numel(x) > 1 && numel(x) ~= S
    error(['Requested ', num2str(S), ' sources. Given ', num2str(numel(x)), ' ', what, '.']);
end
if iscell(x), x = cell2mat(x(:)'); end
if numel(x) == 1, x = repmat(x, 1, S); end
if clamp, x = max(x, 0); end
v = double(x);
end
