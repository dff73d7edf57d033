function aca = b2a(bc)
% B2A - drop-in replacement of the reference's b2a.m (minimum-phase alpha of a beta polynomial) on the MI355X.
[ar, ai] = mbfir_slr_mex(0, bc);
aca = ar + 1i*ai;
