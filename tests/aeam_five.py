"""A five-element AEAM potential file made of the blocks of the bundled two-element file (test data generator)."""
import numpy as np


def write_five_element_file(path, pot):
    """a 5-element potential file (3 non-angular, 2 angular) made of the blocks of AlSi.aeam: Ala, Alb, Alc behave as
    Al, Sia and Sib as Si -- so a 5-type system must reproduce the 2-type system atom for atom"""
    lines = open(pot).read().split("\n")
    head, body = lines[:11], lines[18:]
    vals = np.array(" ".join(body).split(), dtype=float)
    n = 10000
    assert len(vals) == 9 * n
    F = [vals[0:n], vals[n:2 * n]]
    rhor = {(a, b): vals[(2 + 2 * a + b) * n:(3 + 2 * a + b) * n] for a in range(2) for b in range(2)}
    z2r = {(0, 0): vals[6 * n:7 * n], (1, 0): vals[7 * n:8 * n], (1, 1): vals[8 * n:9 * n]}
    cls = [0, 0, 0, 1, 1]
    names = ["Ala", "Alb", "Alc", "Sia", "Sib"]
    el = [lines[12], lines[13]]                      # nrho drho mass of Al, Si
    pr = {(0, 0): lines[14], (0, 1): lines[15], (1, 0): lines[16], (1, 1): lines[17]}
    out = head + ["5 3 2 " + " ".join(names)]
    out += [" ".join(el[c].split()[:3]) + " " + nm for c, nm in zip(cls, names)]
    out += [" ".join(pr[(cls[i], cls[j])].split()[:3]) for i in range(5) for j in range(5)]

    def block(v):
        return [" ".join("%.16e" % x for x in v[k:k + 5]) for k in range(0, len(v), 5)]
    for c in cls:
        out += block(F[c])
    for i in range(5):
        for j in range(5):
            out += block(rhor[(cls[i], cls[j])])
    for i in range(5):
        for j in range(i + 1):
            out += block(z2r[(max(cls[i], cls[j]), min(cls[i], cls[j]))])
    open(path, "w").write("\n".join(out) + "\n")
