import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbfir
for n in (1024,):
    c, g = mbfir.time_kernels(n=n, m=16394, nt=1023, reps=30)
    print("n=%d chol+inv %.3f ms ; k_gram(16394x1023) %.3f ms = %.1f TFLOP/s" % (n, c, g, 16394 * 1023 * 1024 / g / 1e9))
