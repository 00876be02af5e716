"""ctypes front end of oracle/kge_oracle_c.c (multi-threaded C restatement of the bench workload; TEST
INFRASTRUCTURE -- only tests/ and bench.py's cpu_baseline leg use it)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "kge_oracle_c.c")
LIB = os.path.join(HERE, "_build", "libkge_oracle_c.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"], check=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        build()  # no-op when the library is newer than the source
        lib = ctypes.CDLL(LIB)
        lib.kgec_threads.restype = ctypes.c_int
        lib.kgec_set_threads.argtypes = [ctypes.c_int]
        lib.kgec_transe_adam_step.restype = ctypes.c_float
        lib.kgec_transe_adam_step.argtypes = ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 6 +
                                              [ctypes.c_int64, ctypes.c_float, ctypes.c_int64] + [ctypes.c_void_p] * 6)
        lib.kgec_transe_eval.restype = None
        lib.kgec_transe_eval.argtypes = ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 5)
        _lib = lib
    return _lib


def threads():
    return load().kgec_threads()


def set_threads(n):
    load().kgec_set_threads(int(n))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class TransEAdam:
    """State of a dense-Adam TransE run; tables are updated in place."""

    def __init__(self, ent, rel, l1_flag, margin, lr):
        self.ent = np.ascontiguousarray(ent, np.float32)
        self.rel = np.ascontiguousarray(rel, np.float32)
        self.l1, self.margin, self.lr, self.step = int(bool(l1_flag)), float(margin), float(lr), 0
        z = lambda a: np.zeros_like(a)
        self.g_ent, self.g_rel = z(self.ent), z(self.rel)
        self.m_ent, self.v_ent, self.m_rel, self.v_rel = z(self.ent), z(self.ent), z(self.rel), z(self.rel)

    def train_step(self, ph, pr, pt, nh, nr, nt):
        ids = [np.ascontiguousarray(a, np.int64) for a in (ph, pr, pt, nh, nr, nt)]
        self.step += 1
        return float(load().kgec_transe_adam_step(_p(self.ent), _p(self.rel), self.ent.shape[0], self.rel.shape[0],
                                                  self.ent.shape[1], self.l1, self.margin, *[_p(a) for a in ids],
                                                  len(ids[0]), self.lr, self.step, _p(self.g_ent), _p(self.g_rel),
                                                  _p(self.m_ent), _p(self.v_ent), _p(self.m_rel), _p(self.v_rel)))


def transe_eval(ent, rel, l1_flag, triples, tail_off, tail_ids, head_off, head_ids):
    ent = np.ascontiguousarray(ent, np.float32)
    rel = np.ascontiguousarray(rel, np.float32)
    trip = np.ascontiguousarray(triples, np.int64)
    n = trip.shape[0]
    ranks = np.zeros((4, n), dtype=np.int32)
    arrs = [None if a is None else np.ascontiguousarray(a, dt) for a, dt in
            ((tail_off, np.int64), (tail_ids, np.int32), (head_off, np.int64), (head_ids, np.int32))]
    load().kgec_transe_eval(_p(ent), _p(rel), ent.shape[0], ent.shape[1], int(bool(l1_flag)), _p(trip), n,
                            *[_p(a) for a in arrs], _p(ranks))
    return ranks
