function rf = ab2rf(ac, bc)
% AB2RF - drop-in replacement of the reference's ab2rf.m (inverse SLR transform, n <= 2048) on the MI355X.
[rr, ri] = mbfir_slr_mex(1, ac, bc);
rf = rr + 1i*ri;
