function [h, status] = fir_qp_cvx(n, f, a, d, k, obj, dbg)
% FIR_QP_CVX - drop-in replacement of the reference's fir_qp_cvx.m (n x 1 taps).
if nargin < 4,  error('not enough input');  end;
if nargin == 4, k = 100; end;
if nargin <= 5, obj = 0; end;
if numel(obj) ~= 1 && numel(obj) ~= 2, error('invalid input of obj'); end;
[hr, hi, rc] = mbfir_mex(1, n, f, a, d, k, obj);
if rc == 0, h = hr + 1i*hi; status = 'Solved'; else h = []; status = 'Failed'; end
