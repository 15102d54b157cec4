"""child of tests/test_gpu_parity.py::test_exported_yama_dies_like_the_reference: calls the exported yama() (reference signature,
mz_yama.h:22) of <lib> with band case <k> and lets it die: the message goes to stderr, the exit code is the process's.
    python tests/tools/yama_fatal_child.py <lib.so> <case>"""
import ctypes as C
import sys
import numpy as np
L = C.CDLL(sys.argv[1])
k = int(sys.argv[2])
C.c_char_p.in_dll(L, "argv0").value = b"/some/where/multiz"          # util.c:4 / print_argv0(): the message's prefix
M, N, K, Lr = 60, 64, 2, 2
rng = np.random.default_rng(4)
A = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(M, K))].copy()
B = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(N, Lr))].copy()
i = np.arange(M + 1)
LB = np.maximum(i * N // M - 20, 0).astype(np.int32); RB = np.minimum(i * N // M + 20, N).astype(np.int32)
if k == 0: LB[0] = 1                                    # "LB and RB not terminated properly"
elif k == 1: RB[M] = N - 1                              # the same, other end
elif k == 2: RB[20] = LB[20] + 3                        # "RB[20] - LB[20] < 10, ..."
elif k == 3: LB[40] = LB[39] - 1                        # "LB not monotonic"
elif k == 4: RB[30] = RB[29] - 1                        # "RB not monotonic"
elif k == 5: RB[5] = LB[5] + 2; LB[7] = LB[6] - 1       # two faults: the first in row order wins
pa = (C.c_void_p * (M + 1))(); pb = (C.c_void_p * (N + 1))()
for r in range(1, M + 1): pa[r] = A.ctypes.data + (r - 1) * K
for c in range(1, N + 1): pb[c] = B.ctypes.data + (c - 1) * Lr
oal = C.c_void_p(); om = C.c_int(0)
L.yama.restype = None
L.yama(pa, K, M, pb, Lr, N, C.c_void_p(LB.ctypes.data), C.c_void_p(RB.ctypes.data), C.byref(oal), C.byref(om))
print("yama returned", om.value)
