import ctypes, os, numpy as np
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_lds2.so"))
iters = 3000
ms = (ctypes.c_float * 10)()
L.ubench2(iters, ms)
names = ["0 const operands", "1 operands from reads", "2 +barrier only", "3 reads->operands + barrier", "4 swizzled addr",
         "5 swizzle + operands", "7 swizzle+operands+barrier", "15 + setprio", "8 setprio only", "0 again"]
for n, m in zip(names, ms[:]):
    print(f"{n:32s} {m:8.2f} ms  {2*32*32*16*32*iters*8*256/(m*1e-3)/1e12:7.0f} TF")
