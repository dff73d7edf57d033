function [mx, my, mz] = blochH(b1, gr, tp, t1, t2, df, dp, mode, mx, my, mz)
% [mx,my,mz] = blochH(b1,gr,tp,t1,t2,df,dp,mode,mx,my,mz) -- drop-in for the MEX bloch_simulation/blochH.c
% (documented in bloch_simulation/bloch.m:1-44; called by sim_rf_spectral.m:63-78): the simulation runs on the GPU
% through mbfir_bloch (gamma = 26754 rad/s/G, blochH.c:6).  Output shapes follow the reference gateway's final reshape
% (blochC.c:878-903): ntout x npos x nfreq when all three exceed one, else ntout x (npos*nfreq) or npos x nfreq.
if nargin < 8, mode = 0; end
if nargin < 11
    [mx, my, mz] = mbfir_bloch_mex(26754, b1, gr, tp, t1, t2, df, dp, mode);
else
    [mx, my, mz] = mbfir_bloch_mex(26754, b1, gr, tp, t1, t2, df, dp, mode, mx, my, mz);
end
ntout = size(mx, 1);
nf = numel(df);
npos = numel(mx) / (ntout * nf);
if ntout > 1 && nf > 1 && npos > 1
    mx = reshape(mx, [ntout, npos, nf]); my = reshape(my, [ntout, npos, nf]); mz = reshape(mz, [ntout, npos, nf]);
elseif ntout == 1
    mx = reshape(mx, [npos, nf]); my = reshape(my, [npos, nf]); mz = reshape(mz, [npos, nf]);
end
