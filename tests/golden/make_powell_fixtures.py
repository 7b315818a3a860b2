"""Fixture for A11's scalar pieces: scipy.optimize's `bracket` and `Brent` on a handful of scalar functions -> tests/golden/powell_scipy.json.

powell.pas says it "is taken from scipy" (an older optimize.py); tests/test_oracle_pins.py holds the oracle's restatement of its Bracket and
Brent (oracle/tm_oracle.c) to what scipy itself computes on the same functions.  Only derived numbers are committed; scipy is needed to REMAKE
the fixture, not to run the test.

Known differences between powell.pas and today's scipy:
  * constants: scipy works with the rounded `_gold = 1.618034` and `_cg = 0.3819660`; powell.pas computes Gold = (1 + Sqrt(5)) / 2 and
    CG = (3 - Sqrt(5)) / 2 (powell.pas:60, :151).  The fixture is recorded with scipy's own source, that literal replaced, and the instance attribute set to the
    exact values (scipy's CODE, powell.pas' constants): Bracket then agrees bit for bit, points and number of function evaluations.
  * Brent evaluates f at the bracket's middle point once more (powell.pas:262) where scipy keeps the bracket's value: one evaluation more.
  * tolerance: powell.pas works with an ABSOLUTE xtol (tol1 = xtol); scipy with tol1 = tol * |x| + _mintol.  The fixture is recorded with tol = 0 and
    the class attribute _mintol = xtol, which makes scipy's tol1 the same absolute number.
  * the parabolic step is pulled back from the bracket's ends when it comes within xtol of them in powell.pas, within tol2 = 2 * tol1 in
    scipy; and powell.pas stops on `<=` where scipy stops on `<`.  Iterates therefore agree until a step lands between xtol and 2 xtol from
    an end (near convergence); the test asserts equal iterates only where the fixture's `exact` flag says this run saw no such step, and
    closeness (|x - x_scipy| <= 4 xtol, f no worse than scipy's by more than the function's change over that distance) everywhere.
Run:  python tests/golden/make_powell_fixtures.py
"""
import ctypes
import json
import os

import scipy
from scipy.optimize._optimize import Brent

HERE = os.path.dirname(os.path.abspath(__file__))


def fn(i, x):  # the same expressions, in the same order, as tmo_test_scalar_fn (oracle/tm_oracle.c)
    if i == 0:
        return (x - 2.0) * (x - 2.0) + 1.0
    if i == 1:
        return x * x * x * x - 3.0 * x * x * x + 2.0
    if i == 2:
        return (x + 1.5) * (x + 1.5) * (x - 0.3) * (x - 0.3) + 0.1 * x
    if i == 3:
        return abs(x - 0.7) + 0.01 * x * x
    if i == 4:
        return x * x / (1.0 + x * x) - 0.2 * x
    if i == 5:
        return -1.0 / (1.0 + (x - 3.0) * (x - 3.0))
    if i == 6:
        return (x - 0.25) * (x - 0.25) * (x - 0.25) * (x - 0.25) + 0.5 * (x - 0.25) * (x - 0.25)
    return 1e3 * (x + 40.0) * (x + 40.0) - 7.0


def main():
    import inspect
    import math
    import scipy.optimize._optimize as so
    # scipy's own code with powell.pas' constants: `_gold` is a local of bracket(), so its source is taken as it is, the one literal
    # replaced, and put back into scipy's module (where Brent finds it); `_cg` and `_mintol` are instance attributes, set below
    src = inspect.getsource(so.bracket)
    assert "_gold = 1.618034" in src
    ns = so.__dict__
    ns["_tm_sqrt"] = math.sqrt
    exec(src.replace("_gold = 1.618034", "_gold = (1.0 + _tm_sqrt(5.0)) / 2.0"), ns)  # powell.pas:60
    bracket_fn = ns["bracket"]
    cg = (3.0 - math.sqrt(5.0)) / 2.0  # powell.pas:151
    lib = ctypes.CDLL(os.path.join(HERE, "..", "..", "oracle", "libtm_oracle.so"))  # only to set the `exact` flags
    lib.tmo_test_brent.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    out = {"scipy": scipy.__version__, "bracket": [], "brent": []}
    for i in range(8):
        for xa, xb in ((0.0, 1.0), (1.0, 0.0), (-3.0, 5.0), (10.0, 10.5)):
            try:
                a, b, c, fa, fb, fc, calls = bracket_fn(lambda x: fn(i, x), xa, xb)
            except Exception:  # noqa: BLE001 -- scipy validates the result (BracketError); powell.pas does not: such starts are left out
                continue
            out["bracket"].append({"fn": i, "xa": xa, "xb": xb, "ends": sorted([float(a), float(c)]), "mid": float(b), "calls": int(calls)})
        for xtol in (0.1, 1e-3, 1e-8):
            br = Brent(lambda x: fn(i, x), tol=0.0, maxiter=100, full_output=True)
            br._mintol = xtol
            br._cg = cg
            br.set_bracket((0.0, 1.0))  # two points: scipy brackets from them, as powell.pas' Brent does with (0, 1)
            try:
                br.optimize()
            except Exception:  # noqa: BLE001
                continue
            x, fx, it, calls = br.get_result(full_output=True)
            res = (ctypes.c_double * 3)()
            lib.tmo_test_brent(i, xtol, 100, res)
            out["brent"].append({"fn": i, "xtol": xtol, "x": float(x), "fx": float(fx), "iter": int(it), "calls": int(calls),
                                 "exact": bool(res[0] == float(x) and int(res[2]) == int(calls) + 1)})
    with open(os.path.join(HERE, "powell_scipy.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(out["bracket"]), "bracket cases,", len(out["brent"]), "brent cases,", sum(c["exact"] for c in out["brent"]), "with identical iterates")


if __name__ == "__main__":
    main()
