"""ORACLE tooling (test infrastructure): ctypes driver for oracle/_ref/libcasadi_ref.so — the reference's own
CasADi-generated C (MHPC/MHPC-Trajopt/CasadiGen/source/*.cpp, HKDMPC/HKD-TrajOpt/CasadiGen/source/*.cpp)
compiled by `make -C oracle ref`.  Scatters the CCS outputs to dense column-major exactly like the
reference's common/casadi_interface.cpp:49-70."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "_ref", "libcasadi_ref.so")
LL = C.c_longlong


def available():
    return os.path.exists(PATH)


class Ref:
    def __init__(self):
        self.lib = C.CDLL(PATH)

    def call(self, name, *args):
        f = getattr(self.lib, name)
        sp = getattr(self.lib, name + "_sparsity_out")
        sp.restype = C.POINTER(LL)
        sp.argtypes = [LL]
        nout = getattr(self.lib, name + "_n_out")
        nout.restype = LL
        n_out = nout()
        ins = [np.ascontiguousarray(a, dtype=np.float64).ravel() for a in args]
        argp = (C.POINTER(C.c_double) * len(ins))(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in ins])
        pats, bufs = [], []
        for i in range(n_out):
            p = sp(i)
            nrow, ncol = p[0], p[1]
            colind = [p[2 + j] for j in range(ncol + 1)]
            nnz = colind[-1]
            rows = [p[2 + ncol + 1 + j] for j in range(nnz)]
            pats.append((nrow, ncol, colind, rows))
            bufs.append(np.zeros(max(nnz, 1)))
        resp = (C.POINTER(C.c_double) * n_out)(*[b.ctypes.data_as(C.POINTER(C.c_double)) for b in bufs])
        iw = (LL * 1)()
        w = (C.c_double * 1)()
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        f(argp, resp, iw, w, 0)
        outs = []
        for (nrow, ncol, colind, rows), b in zip(pats, bufs):
            M = np.zeros((nrow, ncol))
            for c in range(ncol):
                for z in range(colind[c], colind[c + 1]):
                    M[rows[z], c] = b[z]
            outs.append(M)
        return outs
