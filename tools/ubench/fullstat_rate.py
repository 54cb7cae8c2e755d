"""Rate of the full-stat feature pass alone (diagnostic build, -DPAREBEN_DIAG): 256 workgroups, each with
its own Sigma, K = 10000.  usage: fullstat_rate.py <lib.so> [M ...]"""
import ctypes as C, sys
L = C.CDLL(sys.argv[1])
L.pareben_diag_fullstat.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
Ms = [int(v) for v in sys.argv[2:]] or [64, 128, 200, 256, 300, 400, 512, 768]
K = 10000
for M in Ms:
    ms = C.c_double(0)
    rc = L.pareben_diag_fullstat(M, K, 256, 4, C.byref(ms))
    alg = 2.0 * K * M * M
    print("M=%4d  %8.3f ms/pass  algorithmic %.1f GFLOP/s per CU (MFMA work = half; FP64 matrix peak 307 per CU)  rc=%d"
          % (M, ms.value, alg / (ms.value * 1e-3) / 1e9, rc), flush=True)
