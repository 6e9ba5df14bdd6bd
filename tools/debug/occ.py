import ctypes, os
lib = ctypes.CDLL(os.environ["SRK_LIB_PATH"])
a = (ctypes.c_int * 4)()
print("rc", lib.srk_debug_occupancy(a), "blocks/CU: wino4h(256 thr) =", a[0], " wino4(512 thr) =", a[1], " LDS per CU", a[2], " LDS per block", a[3])
