function [h, status] = fir_linprog(n, f, a, d, h0, dbg)
% FIR_LINPROG - drop-in replacement of the reference's ss/fir_linprog.m (n x 1 taps).  h0 (warm start
% of the reference's active-set linprog) is accepted and ignored by the interior-point solver.
[hr, hi, rc] = mbfir_mex(2, n, f, a, d);
if rc == 3, warning('n odd and frequency spec 1 at fs/2'); end
if rc == 0, h = hr + 1i*hi; if all(hi == 0), h = hr; end; status = 'Solved'; else h = []; status = 'Failed'; end
