"""Float comparisons of the GPU tests with their measured error on record.

north_star asks for "within 1e-5 relative" on float outputs.  Two readings are measured for every comparison that goes
through `check_close`: the max-norm error (largest |got - want| over the largest |want|) and the element-relative error over
the entries that are not small (|want| > floor * max|want|; below that an fp32 result carries cancellation noise relative to
the LARGE terms it was summed from, not to itself).  With DEFTET_TOLERANCE_REPORT=<file.jsonl> every call appends its two
measured values, so that the asserted bounds can be kept at a small multiple of what the hardware actually delivers
(profiles/r05_tolerances.json is such a run at the BASELINE sizes).
"""
import json
import os

import numpy as np


def _np64(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float64)


def check_close(name, got, want, maxnorm, elem_rel=None, floor=1e-3, mask=None):
    got, want = _np64(got), _np64(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if mask is not None:
        m = np.asarray(mask.detach().cpu().numpy() if hasattr(mask, "detach") else mask, dtype=bool)
        got, want = got[m], want[m]
    scale = float(np.abs(want).max()) if want.size else 0.0
    err = np.abs(got - want)
    mn = float(err.max() / scale) if scale > 0 else float(err.max() if err.size else 0.0)
    big = np.abs(want) > floor * scale
    er = float((err[big] / np.abs(want[big])).max()) if big.any() else 0.0
    path = os.environ.get("DEFTET_TOLERANCE_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"name": name, "n": int(want.size), "scale": scale, "maxnorm_err": mn, "elem_rel_err": er,
                                "elem_rel_floor": floor, "asserted_maxnorm": maxnorm, "asserted_elem_rel": elem_rel}) + "\n")
    assert mn <= maxnorm, "%s: max-norm error %.3g > %.3g" % (name, mn, maxnorm)
    if elem_rel is not None:
        assert er <= elem_rel, "%s: element-relative error %.3g > %.3g (entries above %g of the maximum)" % (name, er, elem_rel, floor)
    return mn, er
