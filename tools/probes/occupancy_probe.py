import ctypes, torch, os
lib = ctypes.CDLL(os.environ["DEFTET_HIP_LIB"])
out = (ctypes.c_int * 8)()
torch.zeros(1, device="cuda")
print("rc", lib.deftet_debug_occupancy(out), "wave/pair/slab WGs per CU:", out[0], out[1], out[2], "| wave: static LDS", out[3], "regs", out[4], "maxDyn", out[5], "| device LDS per CU", out[6], "per block", out[7])
