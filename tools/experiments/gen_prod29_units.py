#!/usr/bin/env python3
"""prod29::UNIT_TAB of hdn_amd/csrc/xcorr.hip: the 125 vertical 5x1 strips (row block b, column j) of a 5x5 (x) 29x29 plane
assigned to 4 groups of 32 lanes (2 rounds x 2 half waves) so that the lanes of a group read distinct LDS banks from a plane whose
rows are padded to 36 floats: bank = (20 b + 4 r + j + v) mod 32, i.e. distinct (20 b + j) mod 32 within a group.  The values 0, 8
and 20 occur five times among the 125 strips, so three groups carry one 2-way conflict each."""
units = [(b, j) for b in range(5) for j in range(25)]
phi = lambda u: (20 * u[0] + u[1]) % 32
groups, left = [dict() for _ in range(4)], []
for u in sorted(units, key=lambda u: (phi(u), u)):
    cands = [g for g in groups if phi(u) not in g and len(g) < 32]
    if cands:
        min(cands, key=len)[phi(u)] = u
    else:
        left.append(u)
slots = [[g.get(p) for p in range(32)] for g in groups]
for u in left:
    row = next(r for r in slots if None in r)
    row[row.index(None)] = u
flat = [x for row in slots for x in row]
assert sorted(x for x in flat if x) == sorted(units)
print(", ".join(str(255 if x is None else x[0] * 32 + x[1]) for x in flat))
