import ctypes as C, sys
L = C.CDLL("librccl.so.1")
comms = (C.c_void_p * 2)()
devs = (C.c_int * 2)(0, 0)
L.ncclCommInitAll.restype = C.c_int
rc = L.ncclCommInitAll(comms, 2, devs)
L.ncclGetErrorString.restype = C.c_char_p
print("ncclCommInitAll([0,0]) ->", rc, L.ncclGetErrorString(rc))
