function [v, usediters] = nmfx_projfunc(s, k1, k2, nn)
% Drop-in for projfunc(s, k1, k2, nn) of the NMF Toolbox (Hoyer's projection onto sum(abs(v)) = k1, sum(v.^2) = k2, v >= 0 when nn):
% float64 end to end on an AMD MI355X (libnmfx).  A matrix s projects every column independently (extension; usediters is then a
% vector).  SOURCE ONLY, see nmfx_nmf.m.
[v, it] = nmfx_mex('projfunc', double(s), k1, k2, double(nn ~= 0));
usediters = double(it);
end
