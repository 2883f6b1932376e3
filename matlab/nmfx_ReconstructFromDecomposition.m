function V_hat = nmfx_ReconstructFromDecomposition(W, H)
% Drop-in for ReconstructFromDecomposition(W, H) of the NMF Toolbox: W*H for a matrix W, sum_t W(:,:,t) * [zeros(K, t-1) H(:, 1:n-t+1)]
% for an m x K x T tensor, on an AMD MI355X (libnmfx, shift views instead of padded copies).  SOURCE ONLY, see nmfx_nmf.m.
V_hat = nmfx_mex('reconstruct', double(W), double(H));
end
