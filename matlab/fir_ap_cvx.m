function [h, status] = fir_ap_cvx(n, f, a, d, obj, Peak, dbg)
% FIR_AP_CVX - drop-in replacement of the reference's fir_ap_cvx.m (same signature, same
% return convention) that runs the MI355X HIP solver through mbfir_mex instead of CVX.
%   h: 1 x n taps (row), status: 'Solved' | 'Failed' (h = [] on failure)
if nargin < 4,   error('not enough input');  end;
if nargin <= 4,  obj = 0;     end;
if nargin <= 5,  Peak = 1e-3; end;
if obj < 0,      error('invalid input of obj'); end;
[hr, hi, rc] = mbfir_mex(0, n, f, a, d, obj, Peak);
if rc == 0, h = (hr + 1i*hi).'; status = 'Solved'; else h = []; status = 'Failed'; end
