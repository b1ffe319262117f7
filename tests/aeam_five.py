"""AEAM potential files with other element counts, made of the blocks of the bundled two-element file (test data
generator): element k of the new file behaves as Al (cls[k] = 0) or as Si (cls[k] = 1), so a system labelled with the
new elements must reproduce the two-type system atom for atom."""
import numpy as np


def write_relabelled_file(path, pot, cls, names):
    """cls: 0 (Al, non-angular) or 1 (Si, angular) per new element, the non-angular ones first (file format)"""
    assert len(cls) == len(names) and sorted(cls) == list(cls)
    ne, nnon = len(cls), list(cls).count(0)
    lines = open(pot).read().split("\n")
    head, body = lines[:11], lines[18:]
    vals = np.array(" ".join(body).split(), dtype=float)
    n = 10000
    assert len(vals) == 9 * n
    F = [vals[0:n], vals[n:2 * n]]
    rhor = {(a, b): vals[(2 + 2 * a + b) * n:(3 + 2 * a + b) * n] for a in range(2) for b in range(2)}
    z2r = {(0, 0): vals[6 * n:7 * n], (1, 0): vals[7 * n:8 * n], (1, 1): vals[8 * n:9 * n]}
    el = [lines[12], lines[13]]                      # nrho drho mass of Al, Si
    pr = {(0, 0): lines[14], (0, 1): lines[15], (1, 0): lines[16], (1, 1): lines[17]}
    out = head + ["%d %d %d " % (ne, nnon, ne - nnon) + " ".join(names)]
    out += [" ".join(el[c].split()[:3]) + " " + nm for c, nm in zip(cls, names)]
    out += [" ".join(pr[(cls[i], cls[j])].split()[:3]) for i in range(ne) for j in range(ne)]

    def block(v):
        return [" ".join("%.16e" % x for x in v[k:k + 5]) for k in range(0, len(v), 5)]
    for c in cls:
        out += block(F[c])
    for i in range(ne):
        for j in range(ne):
            out += block(rhor[(cls[i], cls[j])])
    for i in range(ne):
        for j in range(i + 1):
            out += block(z2r[(max(cls[i], cls[j]), min(cls[i], cls[j]))])
    open(path, "w").write("\n".join(out) + "\n")


def write_five_element_file(path, pot):
    """3 non-angular + 2 angular elements: Ala, Alb, Alc behave as Al, Sia and Sib as Si"""
    write_relabelled_file(path, pot, [0, 0, 0, 1, 1], ["Ala", "Alb", "Alc", "Sia", "Sib"])
