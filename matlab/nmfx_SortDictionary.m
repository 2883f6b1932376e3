function [W_sorted, H_sorted] = nmfx_SortDictionary(W, H)
% Drop-in for SortDictionary(W, H) of the NMF Toolbox: basis elements ordered by increasing centre of mass (stable for ties), H's rows
% permuted alike; on an AMD MI355X (libnmfx).  SOURCE ONLY, see nmfx_nmf.m.
if nargin < 2
    W_sorted = nmfx_mex('sortdictionary', double(W), []);
else
    [W_sorted, H_sorted] = nmfx_mex('sortdictionary', double(W), double(H));
end
end
