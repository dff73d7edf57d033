function [a, b] = abrm(rf, g, x)
% ABRM - drop-in replacement of rf_tools/abrm.m for 1-D position vectors: Cayley-Klein parameters of the pulse.
if nargin == 2, x = g; g = []; end
[ar, ai, br, bi] = mbfir_slr_mex(3, rf, g, x, 0);
a = (ar + 1i*ai).';  b = (br + 1i*bi).';
if nargout == 1, a = [a b]; end
