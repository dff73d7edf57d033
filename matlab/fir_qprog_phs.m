function [h, status] = fir_qprog_phs(n, f, ac, dc, x0, dbg)
% FIR_QPROG_PHS - drop-in replacement of the reference's ss/fir_qprog_phs.m (n x 1 taps).
[hr, hi, rc] = mbfir_mex(3, n, f, ac, dc);
if rc == 3, warning('n odd and frequency spec non-zero at fs/2'); end
if rc == 0, h = hr + 1i*hi; status = 'Solved'; else h = []; status = 'Failed'; end
