"""The RCCL test double (tests/native/fake_rccl.cpp) against its own contract, on the CPU: the multi-rank GPU tests rely
on it to FAIL where a wrong exchange schedule would hang on the wire, so its matching rules are tested by themselves
(host buffers, MDP_FAKE_RCCL_HOSTMEM=1; ranks are threads).  Runs in a child process: the double reads its switches once."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")

CHILD = r'''
import ctypes as C, json, sys, threading
import numpy as np
L = C.CDLL(sys.argv[1])
case = sys.argv[2]
F64, I32, SUM, MAX = 8, 2, 0, 2          # ncclFloat64, ncclInt32, ncclSum, ncclMax (rccl.h)
class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
for f in (L.ncclSend, L.ncclRecv):
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
L.ncclCommDestroy.argtypes = [C.c_void_p]
uid = Uid()
assert L.ncclGetUniqueId(C.byref(uid)) == 0
def p(a): return a.ctypes.data

def ranks(n, body):
    out, err = [None] * n, [None] * n
    def run(r):
        try:
            comm = C.c_void_p()
            rc = L.ncclCommInitRank(C.byref(comm), n, uid, r)
            assert rc == 0, rc
            out[r] = body(r, n, comm)
            L.ncclCommDestroy(comm)
        except BaseException as e:
            err[r] = repr(e)
    th = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    return out, err

def matched(r, n, comm):
    # the shape of mdp_dd_comm_forward_begin: a nested group of ragged sends/recvs plus one flag word per peer
    send = [np.full(3 * (r + q + 1), 100.0 * r + q) for q in range(n)]
    recv = [np.zeros(3 * (r + q + 1)) for q in range(n)]
    flag, flags = np.array([float(r + 1)]), np.zeros(n)
    rcs = [L.ncclGroupStart(), L.ncclGroupStart()]
    for q in range(n):
        rcs.append(L.ncclSend(p(send[q]), send[q].size, F64, q, comm, None))
        rcs.append(L.ncclRecv(p(recv[q]), recv[q].size, F64, q, comm, None))
    rcs.append(L.ncclGroupEnd())
    for q in range(n):
        rcs.append(L.ncclSend(p(flag), 1, F64, q, comm, None))
        rcs.append(L.ncclRecv(p(flags[q:]), 1, F64, q, comm, None))
    rcs.append(L.ncclGroupEnd())
    ok = all(np.all(recv[q] == 100.0 * q + r) for q in range(n)) and list(flags) == [q + 1.0 for q in range(n)]
    cnt, allc = np.arange(n, dtype=np.int32) + 10 * r, np.zeros(n * n, dtype=np.int32)
    rcs.append(L.ncclAllGather(p(cnt), p(allc), n, I32, comm, None))
    ok = ok and list(allc) == [10 * q + k for q in range(n) for k in range(n)]
    v, s, m = np.array([r + 0.5, -r]), np.zeros(2), np.zeros(2)
    rcs.append(L.ncclAllReduce(p(v), p(s), 2, F64, SUM, comm, None))
    rcs.append(L.ncclAllReduce(p(v), p(m), 2, F64, MAX, comm, None))
    ok = ok and s[0] == sum(q + 0.5 for q in range(n)) and m[0] == n - 0.5 and m[1] == 0.0
    return dict(ok=bool(ok), rcs=rcs)

def size_mismatch(r, n, comm):
    a = np.zeros(8)
    if r == 0:
        return L.ncclSend(p(a), 4, F64, 1, comm, None)
    return L.ncclRecv(p(a), 5, F64, 0, comm, None)

def unmatched_recv(r, n, comm):
    a = np.zeros(8)
    if r == 0:
        return L.ncclRecv(p(a), 4, F64, 1, comm, None)      # rank 1 never sends
    return 0

def unmatched_send(r, n, comm):
    a = np.zeros(8)
    if r == 1:
        return L.ncclSend(p(a), 4, F64, 0, comm, None)      # rank 0 never receives
    return 0

def order_mismatch(r, n, comm):
    a, b = np.ones(1), np.zeros(1)
    if r == 0:      # collective first, then the message ...
        rc1 = L.ncclAllReduce(p(a), p(b), 1, F64, SUM, comm, None)
        return [rc1, L.ncclSend(p(a), 1, F64, 1, comm, None)]
    rc1 = L.ncclRecv(p(b), 1, F64, 0, comm, None)             # ... the peer the other way round: a deadlock on the wire
    return [rc1, L.ncclAllReduce(p(a), p(b), 1, F64, SUM, comm, None)]

def collective_mismatch(r, n, comm):
    a, b = np.ones(4), np.zeros(8)
    if r == 0:
        return L.ncclAllReduce(p(a), p(b), 2, F64, SUM, comm, None)
    return L.ncclAllReduce(p(a), p(b), 3, F64, SUM, comm, None)

n = 3 if case == "matched" else 2
out, err = ranks(n, globals()[case])
print(json.dumps(dict(out=out, err=err)))
'''


def _child(case, timeout_s="2"):
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "lammps-plugins_amd"), "rccl-double"], check=True)
    env = dict(os.environ, MDP_FAKE_RCCL_HOSTMEM="1", MDP_FAKE_RCCL_TIMEOUT_S=timeout_s)
    r = subprocess.run([sys.executable, "-c", CHILD, LIB, case], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_matched_schedule_of_three_ranks_delivers_everything():
    got, _ = _child("matched")
    assert got["err"] == [None, None, None]
    for o in got["out"]:
        assert o["ok"] and set(o["rcs"]) == {0}


def test_message_sizes_must_agree():
    got, log = _child("size_mismatch")
    assert got["out"][1] != 0                       # the receiver sees the disagreement ...
    assert "expects 40 bytes" in log and "carries 32" in log
    assert got["out"][0] != 0                       # ... and the sender's operation does not complete either


@pytest.mark.parametrize("case,who,text", [("unmatched_recv", 0, "never matched by a send"),
                                           ("unmatched_send", 1, "never received")])
def test_an_unmatched_operation_times_out_instead_of_passing(case, who, text):
    got, log = _child(case)
    assert got["out"][who] != 0 and text in log


def test_collectives_and_messages_in_different_orders_fail_on_both_ranks():
    got, _ = _child("order_mismatch")
    assert got["out"][0][0] != 0 and got["out"][1][0] != 0


def test_collective_arguments_must_agree():
    got, log = _child("collective_mismatch")
    assert got["out"][0] != 0 and got["out"][1] != 0
    assert "not in one order" in log
