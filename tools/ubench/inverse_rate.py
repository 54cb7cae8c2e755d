"""Rate of the blocked inverse alone (diagnostic build: tools/build_variants.sh diag): 256 workgroups, each on its own SPD
matrix, every form of gm_spd_inverse (gm_dev.h): one pivot block per trip through memory, two per trip, the register form
(falls back to two per trip beyond its size limit).  usage: inverse_rate.py [M ...]   env CAP (default 1000), BLOCKS (256)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, "pareben_amd", "lib", os.environ.get("DIAG_LIB", "libpareben_hip_diag.so")))
L.pareben_diag_inverse.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
Ms = [int(v) for v in sys.argv[1:]] or [20, 48, 64, 96, 133, 160, 200, 256, 304, 320, 400, 512, 640, 800]
cap = int(os.environ.get("CAP", "1000"))
blocks = int(os.environ.get("BLOCKS", "256"))
print("us per inversion (pivot sweeps / panel product inside): one block per trip | two per trip | register form;  matrix-pipe time; same bits")
for M in Ms:
    reps = max(4, min(200, int(2e4 / M)))
    nT = (M + 15) // 16
    mfma_us = nT * (nT + (nT - 1) * nT // 2) * 4 * 64 / 4 / 2.4e3
    row = []
    for mode in (0, 1, 3):
        ms = C.c_double(0); ph = (C.c_longlong * 24)(); chk = (C.c_double * blocks)()
        rc = L.pareben_diag_inverse(M, max(cap, M), blocks, reps, mode, C.byref(ms), ph, chk)
        row.append((ms.value * 1e3, ph[14] / 100.0 / reps, ph[15] / 100.0 / reps, list(chk), rc, list(ph)))
    same = row[0][3] == row[1][3] == row[2][3]
    print("M=%4d  " % M + " | ".join("%8.1f (%6.1f %6.1f)" % r[:3] for r in row) + "  matrix-pipe %7.1f  same bits %s chk %.9g rc %s"
          % (mfma_us, same, row[0][3][0], [r[4] for r in row]), flush=True)
    if os.environ.get("DETAIL"):
        print("        register form: load %.1f  export %.1f  pivot(all) %.1f  panel %.1f  own tiles %.1f  store %.1f us" % tuple(row[2][5][k] / 100.0 / reps for k in (16, 17, 14, 15, 20, 21)))
