function [W, H, Z, A, cost] = nmfx_constrainednmf(V, labels, num_basis_elems, config)
% nmfx_constrainednmf  Drop-in for constrainednmf(V, labels, num_basis_elems, config) of the NMF Toolbox (V ~ W*Z*A with the 0/1 label
% matrix A), computed on an AMD MI355X by libnmfx.  SOURCE ONLY (never run: no MATLAB in the build image; the gateway underneath is
% exercised by tests/test_mex_gateway.py).  Rename to constrainednmf.m (ahead of the toolbox on the path) to drop in.
% The label bookkeeping stays on the host as in the toolbox: classes renumbered 1..C, unlabelled samples (label -1) first, samples of
% a class contiguous.  The library never sees A: it gets the column ranges of A's non-zeros over the label-sorted samples (segment c
% = samples seg(c)+1 .. seg(c+1) share column c of Z).  config.Z_init (extension) replaces the toolbox's internal rand() for Z.
if nargin < 4, config = struct; end
[m, n] = size(V);
K = num_basis_elems;
assert(length(labels) == n, ['Length of the label vector not equal to number of samples. Length of label vector = ', num2str(length(labels)), '; number of samples = ', num2str(n)]);
if ~isfield(config, 'W_init') || isempty(config.W_init), config.W_init = rand(m, K); end
if ~isfield(config, 'W_sparsity') || isempty(config.W_sparsity), config.W_sparsity = 0; end
if ~isfield(config, 'Z_sparsity') || isempty(config.Z_sparsity), config.Z_sparsity = 0; end
if ~isfield(config, 'W_fixed') || isempty(config.W_fixed), config.W_fixed = false; end
if ~isfield(config, 'Z_fixed') || isempty(config.Z_fixed), config.Z_fixed = false; end
if ~isfield(config, 'divergence'), config.divergence = 'euclidean'; end
is_ab = any(strcmp(config.divergence, {'ab_divergence', 'ab'}));
if ~isfield(config, 'alpha') || ~is_ab, config.alpha = 1; end
if ~isfield(config, 'beta') || ~is_ab, config.beta = 1; end
if ~isfield(config, 'maxiter') || config.maxiter <= 0, config.maxiter = 100; end
if ~isfield(config, 'tolerance') || config.tolerance <= 0, config.tolerance = 1e-3; end
if is_ab && config.alpha == 0 && config.beta == 0, error('alpha = 0 and beta = 0 is not supported at this time.'); end
switch config.divergence
    case 'euclidean', dv = 0;
    case {'kl_divergence', 'kl'}, dv = 1;
    case {'is_divergence', 'is'}, dv = 2;
    case {'ab_divergence', 'ab'}, dv = 3;
    otherwise, error(['No update equations defined for cost function with divergence type ', config.divergence]);
end
% classes -> 1..C, unlabelled -> -1, stable ascending sort: unlabelled samples first, then class by class
labels = labels(:);
n_lab = nnz(labels > -1);
[u, ~, lp] = unique(labels);
if n_lab < n
    lp = lp - 1; lp(lp == 0) = -1;
    C = numel(u) - 1;
else
    C = numel(u);
end
[sl, order] = sort(lp, 'ascend');
n_u = n - n_lab;
seg = zeros(1, n_u + C + 1);
seg(1 : n_u + 1) = 0 : n_u;                                 % every unlabelled sample has its own column of Z
for c = 1 : C, seg(n_u + 1 + c) = seg(n_u + c) + nnz(sl(n_u + 1 : end) == c); end
nz = n_u + C;
if ~isfield(config, 'Z_init') || isempty(config.Z_init), config.Z_init = rand(K, nz); end
opts.divergence = dv; opts.alpha = config.alpha; opts.beta = config.beta;
opts.W_sparsity = double(config.W_sparsity); opts.Z_sparsity = double(config.Z_sparsity);
opts.W_fixed = double(logical(config.W_fixed)); opts.Z_fixed = double(logical(config.Z_fixed));
opts.maxiter = config.maxiter; opts.tolerance = config.tolerance;
[W, Hs, cost, Z] = nmfx_mex('constrainednmf', double(V(:, order)), double(config.W_init), double(config.Z_init), int64(seg), opts);
% back to the caller's sample order: sample order(s) sits in segment zcol(s)
zcol = zeros(1, n);
for c = 1 : nz, zcol(seg(c) + 1 : seg(c + 1)) = c; end
A = zeros(nz, n);
A(sub2ind([nz, n], zcol, order(:)')) = 1;
H = zeros(K, n);
H(:, order) = Hs;
end
