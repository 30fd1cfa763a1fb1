import ctypes, os, numpy as np
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_lds.so"))
iters = 20000
ms = (ctypes.c_float * 20)()
L.ubench(iters, ms)
ms = np.array(ms[:]).reshape(5, 4)
# per SIMD: 2 waves x iters x 4 MFMA ; ideal 32 cycles per MFMA per SIMD
print("rows: R = ds_read_b128 per 4 MFMAs per wave; cols: (dst,acc) = (v,v) (v,a) (a,v) (a,a)")
for R in range(5):
    tf = [2 * 32 * 32 * 16 * 4 * iters * 8 * 256 / (m * 1e-3) / 1e12 for m in ms[R]]
    print(R, " ".join(f"{m:8.2f} ms {t:7.0f} TF" for m, t in zip(ms[R], tf)))
