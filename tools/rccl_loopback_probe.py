#!/usr/bin/env python3
"""rccl_loopback_probe.py — does the real librccl accept a one-rank communicator with a send-to-self / recv-from-self pair in one
group?  (feasibility probe for a loopback check of the multi-channel front's transport; development tool)"""
import ctypes
import sys

import torch

lib = ctypes.CDLL("librccl.so.1")
uid = (ctypes.c_char * 128)()
assert lib.ncclGetUniqueId(uid) == 0
torch.cuda.set_device(0)
comm = ctypes.c_void_p()


class Uid(ctypes.Structure):
    _fields_ = [("b", ctypes.c_char * 128)]


u = Uid.from_buffer_copy(bytes(uid))
lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
r = lib.ncclCommInitRank(ctypes.byref(comm), 1, u, 0)
print("ncclCommInitRank(1 rank):", r)
if r != 0:
    sys.exit(1)
n = 1 << 24
a = torch.arange(n, dtype=torch.float32, device="cuda")
b = torch.zeros(n, dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream()
torch.cuda.synchronize()
lib.ncclSend.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.ncclRecv.argtypes = lib.ncclSend.argtypes
NCCL_INT8 = 0
print("group start", lib.ncclGroupStart())
print("send", lib.ncclSend(a.data_ptr(), 4 * n, NCCL_INT8, 0, comm, stream.cuda_stream))
print("recv", lib.ncclRecv(b.data_ptr(), 4 * n, NCCL_INT8, 0, comm, stream.cuda_stream))
print("group end", lib.ncclGroupEnd())
stream.synchronize()
print("equal:", bool(torch.equal(a, b)))
lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
print("destroy", lib.ncclCommDestroy(comm))
